"""Route B of INTEGRATION.md: ``diff_gaussian_rasterization_w_tof._C`` -- the three pybind-level functions of the
reference's extension (RAST/ext.cpp:15-19, signatures RAST/rasterize_points.h:18-88) -- driven the way the
reference's own Python wrapper drives them (RAST/diff_gaussian_rasterization_w_tof/__init__.py:87-206) and checked
against the oracle."""
import inspect

import numpy as np
import pytest
import torch

import helpers as Hh

# positional parameter names of rasterize_points.h:25-53, :55-88, :90-95
FWD_ARGS = ["background", "means3D", "colors", "phasors", "opacity", "scales", "rotations", "scale_modifier", "cov3D_precomp",
            "viewmatrix", "projmatrix", "tan_fovx", "tan_fovy", "image_height", "image_width", "sh", "sh_p", "degree",
            "campos", "prefiltered", "debug", "near_n", "far_n", "depth_range", "use_view_dependent_phase", "phase_offset",
            "dc_offset"]
BWD_ARGS = ["background", "means3D", "radii", "colors", "phasors", "scales", "rotations", "scale_modifier", "cov3D_precomp",
            "viewmatrix", "projmatrix", "tan_fovx", "tan_fovy", "dL_dout_color", "dL_dout_phasor", "dL_dout_depth",
            "dL_dout_normal", "dL_dout_acc", "dL_dout_entropy", "dL_dout_depth_distortion", "dL_dout_amp_distortion", "sh",
            "sh_p", "degree", "campos", "geomBuffer", "R", "binningBuffer", "imageBuffer", "debug", "near_n", "far_n",
            "depth_range", "use_view_dependent_phase", "phase_offset", "dc_offset"]
VIS_ARGS = ["means3D", "viewmatrix", "projmatrix", "znear", "zfar"]


def test_module_exports_the_pybind_surface():
    from diff_gaussian_rasterization_w_tof import _C
    assert list(inspect.signature(_C.rasterize_gaussians).parameters) == FWD_ARGS and len(FWD_ARGS) == 27
    assert list(inspect.signature(_C.rasterize_gaussians_backward).parameters) == BWD_ARGS and len(BWD_ARGS) == 36
    assert list(inspect.signature(_C.mark_visible).parameters) == VIS_ARGS
    with pytest.raises(RuntimeError, match="HIP device only"):
        z = torch.zeros
        _C.rasterize_gaussians(z(7, 4, 4), z(5, 3), torch.Tensor([]), torch.Tensor([]), z(5, 1), z(5, 3), z(5, 4), 1.0,
                               torch.Tensor([]), torch.eye(4), torch.eye(4), 1.0, 1.0, 4, 4, z(5, 16, 3), z(5, 16, 2), 3, z(3),
                               False, False, 0.01, 100.0, 10.0, True, 0.0, 0.0)


def _call_like_the_reference_wrapper(scene, dev, inputs=None):
    """The statements of _RasterizeGaussians.forward / .backward (reference __init__.py:87-125, 148-190) on top of _C."""
    from diff_gaussian_rasterization_w_tof import _C
    g = dict(scene["gaussians"])
    if inputs:
        g.update(inputs)
    cam, cfg = scene["cam"], scene["cfg"]
    t = lambda a: torch.Tensor([]) if a is None else torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    bg, view, proj, campos = t(scene["bg"]), t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"])
    means3D, opac = t(g["means3D"]), t(g["opacities"])
    colors, phasors = t(g.get("colors_precomp")), t(g.get("phasors_precomp"))
    scales, rots, cov = t(g.get("scales")), t(g.get("rotations")), t(g.get("cov3D_precomp"))
    sh, sh_p = t(g.get("shs")), t(g.get("shs_p"))
    args = (bg, means3D, colors, phasors, opac, scales, rots, 1.0, cov, view, proj, cam["tanfovx"], cam["tanfovy"],
            cfg["H"], cfg["W"], sh, sh_p, cfg["D"], campos, False, False, cam["znear"], cam["zfar"], scene["depth_range"],
            scene["use_view_dependent_phase"], scene["phase_offset"], scene["dc_offset"])
    fw = _C.rasterize_gaussians(*args)
    assert len(fw) == 15
    (num_rendered, color, phasor, depth, normal, acc, entropy, depth_distortion, amp_distortion, pixels, distribution, radii,
     geomBuffer, binningBuffer, imgBuffer) = fw
    gr = {k: t(v) for k, v in scene["grads"].items()}
    zero = lambda x: torch.zeros_like(x)
    bargs = (bg, means3D, radii, colors, phasors, scales, rots, 1.0, cov, view, proj, cam["tanfovx"], cam["tanfovy"],
             gr["color"], gr["phasor"], gr["depth"], zero(normal), gr["acc"], zero(entropy), gr["depth_distortion"],
             zero(amp_distortion), sh, sh_p, cfg["D"], campos, geomBuffer, num_rendered, binningBuffer, imgBuffer, False,
             cam["znear"], cam["zfar"], scene["depth_range"], scene["use_view_dependent_phase"], scene["phase_offset"],
             scene["dc_offset"])
    bw = _C.rasterize_gaussians_backward(*bargs)
    assert len(bw) == 12
    torch.cuda.synchronize()
    return fw, bw


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["sh", "precomp_colors_cov3d"])
def test_pybind_level_calls_match_the_oracle(variant):
    from oracle import oracle
    oracle.build()
    dev = torch.device("cuda:0")
    scene = Hh.small_scene(P=1500, W=96, H=64, seed=5)
    inputs = None
    if variant == "precomp_colors_cov3d":
        rng = np.random.default_rng(3)
        P = scene["cfg"]["P"]
        f0, _ = Hh.run_oracle(oracle, scene, backward=False)
        inputs = dict(shs=None, colors_precomp=rng.random((P, 3)).astype(np.float32), scales=None, rotations=None,
                      cov3D_precomp=np.asarray(f0.geom["cov3D"], np.float32).reshape(P, 6))
    f, b = Hh.run_oracle(oracle, scene, inputs=inputs)
    for rep in range(2):       # the second call takes the hinted one-call forward (binning buffer larger than R)
        fw, bw = _call_like_the_reference_wrapper(scene, dev, inputs)
        R = fw[0]
        assert isinstance(R, int) and R == f.num_rendered
        names = ["color", "phasor", "depth", "normal", "acc", "entropy", "depth_distortion", "amp_distortion", "pixels",
                 "distribution"]
        H, W, P = scene["cfg"]["H"], scene["cfg"]["W"], scene["cfg"]["P"]
        shapes = dict(color=(3, H, W), phasor=(7, H, W), depth=(1, H, W), normal=(3, H, W), acc=(1, H, W), entropy=(1, H, W),
                      depth_distortion=(1, H, W), amp_distortion=(1, H, W), pixels=(P, 1), distribution=(3, H, W))
        for n, got in zip(names, fw[1:11]):
            assert tuple(got.shape) == shapes[n], n
            ref = np.asarray(f[n], np.float64).reshape(shapes[n])
            if n == "pixels":          # integer counts: an alpha on the 1/255 edge flips a count by one
                assert (got.cpu().numpy() != ref).mean() <= 2e-3
                continue
            l1 = float(np.abs(got.cpu().numpy().astype(np.float64) - ref).mean())
            assert l1 < 1e-5 * max(1.0, float(np.abs(ref).max())), (n, l1)
        radii, geomBuffer, binningBuffer, imgBuffer = fw[11:15]
        assert radii.dtype == torch.int32 and (radii.cpu().numpy() == f.radii).all()
        for buf in (geomBuffer, binningBuffer, imgBuffer):
            assert buf.dtype == torch.uint8 and buf.dim() == 1
        assert (binningBuffer.numel() - 256) // 12 >= R
        if rep == 1:
            assert (binningBuffer.numel() - 256) // 12 > R          # sized from the previous frame, not exactly

        (dL_dmeans2D, dL_dcolors, dL_dphasors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dsh_p, dL_dscales,
         dL_drotations, dL_dphase_offset, dL_ddc_offset) = bw
        assert dL_dphasors is None                                   # see _C.py: internal [P,7] array, never formed here
        checks = [("dL_dmeans2D", dL_dmeans2D), ("dL_dcolors", dL_dcolors), ("dL_dopacity", dL_dopacity),
                  ("dL_dmeans3D", dL_dmeans3D), ("dL_dcov3D", dL_dcov3D), ("dL_dsh_p", dL_dsh_p),
                  ("dL_dphase_offset", dL_dphase_offset), ("dL_ddc_offset", dL_ddc_offset)]
        if variant == "sh":
            checks += [("dL_dsh", dL_dsh), ("dL_dscales", dL_dscales), ("dL_drotations", dL_drotations)]
        else:
            assert tuple(dL_dsh.shape) == (P, 0, 3) and not dL_dscales.any() and not dL_drotations.any()
        for n, got in checks:
            ref = np.asarray(b[n])
            assert tuple(got.shape) == ref.shape, (n, tuple(got.shape), ref.shape)
            e, d = Hh.rel_err(ref, got.cpu().numpy())
            assert e < 3e-4, (n, e, d)


@pytest.mark.gpu
def test_pybind_level_mark_visible_and_empty_scene():
    from diff_gaussian_rasterization_w_tof import _C
    from oracle import oracle
    oracle.build()
    dev = torch.device("cuda:0")
    scene = Hh.small_scene(P=3000, W=64, H=48, seed=9, z_lo=0.2, z_hi=8.0)
    cam = scene["cam"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    vis = _C.mark_visible(t(scene["gaussians"]["means3D"]), t(cam["viewmatrix"]), t(cam["projmatrix"]), cam["znear"],
                          cam["zfar"])
    ref = oracle.mark_visible(scene["gaussians"]["means3D"], cam["viewmatrix"], cam["projmatrix"], cam["znear"], cam["zfar"])
    assert vis.dtype == torch.bool and (vis.cpu().numpy() == ref).all() and 0 < ref.sum() < ref.size
    # P == 0: zero images, R = 0 (rasterize_points.cu:104)
    e = torch.Tensor([])
    z = lambda *s: torch.zeros(s, device=dev)
    fw = _C.rasterize_gaussians(t(scene["bg"]), z(0, 3), e, e, z(0, 1), z(0, 3), z(0, 4), 1.0, e, t(cam["viewmatrix"]),
                                t(cam["projmatrix"]), cam["tanfovx"], cam["tanfovy"], 48, 64, z(0, 16, 3), z(0, 16, 2), 3,
                                t(cam["campos"]), False, False, cam["znear"], cam["zfar"], 10.0, True, 0.0, 0.0)
    assert fw[0] == 0 and not fw[1].any() and not fw[2].any() and fw[11].numel() == 0
