"""ORACLE (test infrastructure only) of the fused input assembly -- SURVEY section 8(f) row 1.

Two CPU restatements of the reference's glue, ``gaussian_renderer/__init__.py:81-105``:

* :func:`assemble_eager`: the reference's statements as they stand (torch.zeros + boolean-mask
  assignment, ``rotation_activation = torch.nn.functional.normalize``,
  ``scene/gaussian_model.py:43``), on CPU tensors; autograd of these eager ops is the oracle of
  the backward.
* :func:`assemble_loops`: plain numpy loops over Gaussians (independent of torch indexing
  semantics), used to pin the eager restatement on small cases.

Only tests/, __graft_entry__.smoke() and bench.py's baseline leg may import this module.
"""
import numpy as np
import torch


def assemble_eager(xyz, screenspace_points, opacity, scaling, rotation, rotation_raw, features_color,
                   features_phasor, motion_mask, d_xyz=0.0, d_rot=0.0, d_sh=0.0, d_sh_p=0.0,
                   render_regions=("static", "dynamic")):
    """gaussian_renderer/__init__.py:81-105, names as in the reference (pc.get_xyz -> xyz ...)."""
    means3D = torch.zeros(xyz.shape, device=xyz.device, dtype=xyz.dtype)                      # :81
    means2D = torch.zeros(screenspace_points.shape, device=xyz.device, dtype=xyz.dtype)       # :82
    out_opacity = torch.zeros(opacity.shape, device=xyz.device, dtype=xyz.dtype)              # :83
    scales = torch.zeros(scaling.shape, device=xyz.device, dtype=xyz.dtype)                   # :84
    rotations = torch.zeros(rotation.shape, device=xyz.device, dtype=xyz.dtype)               # :85
    shs = torch.zeros(features_color.shape, device=xyz.device, dtype=xyz.dtype)               # :86
    shs_p = torch.zeros(features_phasor.shape, device=xyz.device, dtype=xyz.dtype)            # :87
    m = motion_mask
    if "static" in render_regions:                                                             # :89-96
        means3D[~m] = xyz[~m]
        means2D[~m] = screenspace_points[~m]
        out_opacity[~m] = opacity[~m]
        scales[~m] = scaling[~m]
        rotations[~m] = rotation[~m]
        shs[~m] = features_color[~m]
        shs_p[~m] = features_phasor[~m]
    if "dynamic" in render_regions:                                                            # :97-104
        means3D[m] = xyz[m] + d_xyz
        means2D[m] = screenspace_points[m]
        out_opacity[m] = opacity[m]
        scales[m] = scaling[m]
        rotations[m] = torch.nn.functional.normalize(rotation_raw[m] + d_rot)
        shs[m] = features_color[m] + d_sh
        shs_p[m] = features_phasor[m] + d_sh_p
    return means3D, means2D, out_opacity, scales, rotations, shs, shs_p


def assemble_loops(xyz, ssp, opacity, scaling, rotation, rotation_raw, fc, fp, mask, d_xyz=0.0, d_rot=0.0,
                   d_sh=0.0, d_sh_p=0.0, render_regions=("static", "dynamic")):
    """Same statements, one Gaussian at a time (numpy float32)."""
    f = np.float32
    P = xyz.shape[0]
    outs = [np.zeros_like(np.asarray(a, f)) for a in (xyz, ssp, opacity, scaling, rotation, fc, fp)]
    row = lambda d, k: (np.asarray(d, f)[k] if isinstance(d, np.ndarray) else f(d))
    k = 0
    for i in range(P):
        if mask[i]:
            if "dynamic" in render_regions:
                outs[0][i] = xyz[i] + row(d_xyz, k)
                outs[1][i] = ssp[i]
                outs[2][i] = opacity[i]
                outs[3][i] = scaling[i]
                q = (rotation_raw[i] + row(d_rot, k)).astype(f)
                n = max(f(np.sqrt(np.sum(q * q, dtype=f))), f(1e-12))
                outs[4][i] = q / n
                outs[5][i] = fc[i] + row(d_sh, k)
                outs[6][i] = fp[i] + row(d_sh_p, k)
            k += 1
        elif "static" in render_regions:
            for o, s in zip(outs, (xyz, ssp, opacity, scaling, rotation, fc, fp)):
                o[i] = s[i]
    return outs
