"""Fused assembly of the rasterizer's per-Gaussian inputs.

Replaces, in the reference's ``render`` (``gaussian_renderer/__init__.py:81-105``), the seven
``torch.zeros`` allocations and the fourteen boolean-mask assignments

    means3D[~motion_mask] = pc.get_xyz[~motion_mask]              # "static" in render_regions
    means3D[motion_mask]  = pc.get_xyz[motion_mask] + d_xyz        # "dynamic" in render_regions
    rotations[motion_mask] = pc.rotation_activation(pc._rotation[motion_mask] + d_rot)
    shs[motion_mask] = pc.get_features_color[motion_mask] + d_sh   ...

by one call with the same results and the same gradients:

    means3D, means2D, opacity, scales, rotations, shs, shs_p = assemble_inputs(
        pc.get_xyz, screenspace_points, pc.get_opacity, pc.get_scaling, pc.get_rotation, pc._rotation,
        pc.get_features_color, pc.get_features_phasor, pc.get_motion_mask,
        d_xyz, d_rot, d_sh, d_sh_p, render_regions)

``d_*`` are the deformation network's outputs for the dynamic Gaussians (one row per True of
``motion_mask``, ``scene/gaussian_model.py:170-174``) or Python floats (``train.py:164``).
The work is done by hand-written gfx950 kernels (``csrc/k_assemble.hip``) through the C ABI in
``include/gftorf_assemble.h``; there is no CPU path.
"""
import ctypes as C
import os

import torch

from . import _lib


# GFT_ASSEMBLE_ALIAS_SH=0: the backward copies the SH rows' gradient into a tensor of its own (rounds 3-5)
_ALIAS_SH_GRADS = os.environ.get("GFT_ASSEMBLE_ALIAS_SH", "1") != "0"


def _is_tensor(x):
    return isinstance(x, torch.Tensor)


def _f32c(t, dev, name):
    if t.device != dev:
        raise RuntimeError("gftorf_amd.assemble_inputs: %s is on %s, expected %s" % (name, t.device, dev))
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _p(t):
    return t.data_ptr() if t is not None and t.numel() else None


class _AssembleInputs(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, screenspace, opacity, scaling, rotation, rotation_raw, feat_color, feat_phasor,
                motion_mask, d_xyz, d_rot, d_sh, d_sh_p, render_static, render_dynamic, validate):
        lib = _lib.load()
        dev = xyz.device
        if dev.type != "cuda":
            raise RuntimeError("gftorf_amd.assemble_inputs runs on a HIP device only (xyz is on %s); "
                               "there is no CPU path" % (dev,))
        P = xyz.size(0)
        M, M_p = feat_color.size(1), feat_phasor.size(1)
        src = [None if t is None else _f32c(t, dev, n)
               for t, n in ((xyz, "xyz"), (screenspace, "screenspace_points"), (opacity, "opacity"),
                            (scaling, "scaling"), (rotation, "rotation"), (rotation_raw, "_rotation"),
                            (feat_color, "features_color"), (feat_phasor, "features_phasor"))]
        xyz_c, ssp_c, op_c, sc_c, rot_c, raw_c, fc_c, fp_c = src
        if any(t is None for t in (xyz_c, ssp_c, op_c, sc_c, raw_c, fc_c, fp_c)):
            raise RuntimeError("assemble_inputs: only `rotation` may be None (the static rows are then normalize(rotation_raw))")
        if motion_mask.dtype != torch.bool or motion_mask.numel() != P:
            raise RuntimeError("motion_mask must be a bool tensor with one entry per Gaussian")
        mask_c = motion_mask.to(dev).contiguous()
        offs = []
        for t, n, shape in ((d_xyz, "d_xyz", (3,)), (d_rot, "d_rot", (4,)), (d_sh, "d_sh", (M, 3)), (d_sh_p, "d_sh_p", (M_p, 2))):
            if _is_tensor(t):
                if tuple(t.shape[1:]) != shape:
                    raise RuntimeError("%s has shape %s, expected (num_dynamic, %s)" % (n, tuple(t.shape), ", ".join(map(str, shape))))
                offs.append(_f32c(t, dev, n))
            else:
                offs.append(float(t))
        # every d_* tensor holds one row per dynamic Gaussian (gaussian_renderer/__init__.py:90-104): one row count,
        # at most P -- checked here without touching the device; the kernels never index past it
        rows = {n: v.size(0) for v, n in zip(offs, ("d_xyz", "d_rot", "d_sh", "d_sh_p")) if _is_tensor(v)}
        if len(set(rows.values())) > 1:
            raise RuntimeError("shape mismatch: the offset tensors disagree on the number of dynamic Gaussians: %s" % (rows,))
        n_off = next(iter(rows.values())) if rows else 0
        if n_off > P:
            raise RuntimeError("shape mismatch: the offset tensors have %d rows for %d Gaussians" % (n_off, P))
        f32 = dict(device=dev, dtype=torch.float32)
        means3D = torch.empty((P, 3), **f32)
        means2D = torch.empty((P, 3), **f32)
        out_op = torch.empty(opacity.shape, **f32)
        scales = torch.empty((P, 3), **f32)
        rotations = torch.empty((P, 4), **f32)
        # An SH tensor whose offset is the scalar 0.0 (train.py:164 before warm-up ends; d_sh_p always -- the network's
        # phasor offsets are zeros, time_utils.py:127, and this package's network hands them over as that scalar) is, with
        # both regions rendered, the feature tensor itself: returned as it is, nothing read, nothing written.
        whole = bool(render_static) and bool(render_dynamic) and _ALIAS_SH_GRADS
        same_fc = whole and not _is_tensor(offs[2]) and offs[2] == 0.0
        same_fp = whole and not _is_tensor(offs[3]) and offs[3] == 0.0
        shs = fc_c if same_fc else torch.empty((P, M, 3), **f32)
        shs_p = fp_c if same_fp else torch.empty((P, M_p, 2), **f32)
        scratch = torch.empty((lib.gft_assemble_scratch_bytes(P),), device=dev, dtype=torch.uint8)

        io = _lib.AssembleIO()
        io.xyz, io.screenspace, io.opacity, io.scaling = _p(xyz_c), _p(ssp_c), _p(op_c), _p(sc_c)
        io.rotation, io.rotation_raw, io.feat_color, io.feat_phasor = _p(rot_c), _p(raw_c), _p(fc_c), _p(fp_c)
        io.motion_mask = _p(mask_c)
        for name, v in zip(("d_xyz", "d_rot", "d_sh", "d_sh_p"), offs):
            if _is_tensor(v):
                setattr(io, name, _p(v))
            else:
                setattr(io, name + "_scalar", v)
        io.num_offset_rows = n_off
        io.scratch = scratch.data_ptr()
        io.out_means3D, io.out_means2D, io.out_opacity = _p(means3D), _p(means2D), _p(out_op)
        io.out_scales, io.out_rotations = _p(scales), _p(rotations)
        io.out_shs, io.out_shs_p = None if same_fc else _p(shs), None if same_fp else _p(shs_p)
        stream = _lib.raw_stream(dev)
        with _lib.on_device(dev):
            _lib.check(lib.gft_assemble_forward(stream, P, M, M_p, int(render_static), int(render_dynamic), C.byref(io)))
            if validate:
                # the reference's masked assignment raises when d_* has another row count
                nd = C.c_int64(0)
                _lib.check(lib.gft_assemble_num_dynamic(stream, P, scratch.data_ptr(), C.byref(nd)))
                for v, n in zip(offs, ("d_xyz", "d_rot", "d_sh", "d_sh_p")):
                    if _is_tensor(v) and v.size(0) != nd.value:
                        raise RuntimeError("shape mismatch: %s has %d rows, motion_mask selects %d Gaussians"
                                           % (n, v.size(0), nd.value))
        ctx.sizes = (P, M, M_p, int(render_static), int(render_dynamic))
        ctx.offs_scalar = [None if _is_tensor(v) else v for v in offs]
        ctx.off_rows = [v.size(0) if _is_tensor(v) else 0 for v in offs]
        ctx.opacity_shape = opacity.shape
        ctx.static_from_raw = rot_c is None
        ctx.save_for_backward(raw_c, offs[1] if _is_tensor(offs[1]) else raw_c.new_empty(0), scratch)
        ctx.set_materialize_grads(False)
        return means3D, means2D, out_op, scales, rotations, shs, shs_p

    @staticmethod
    def backward(ctx, g_means3D, g_means2D, g_opacity, g_scales, g_rotations, g_shs, g_shs_p):
        lib = _lib.load()
        raw_c, d_rot_c, scratch = ctx.saved_tensors
        P, M, M_p, rs, rd = ctx.sizes
        dev = raw_c.device
        need = ctx.needs_input_grad
        f32 = dict(device=dev, dtype=torch.float32)
        new = lambda want, shape: torch.empty(shape, **f32) if want else None
        g_xyz, g_ssp = new(need[0], (P, 3)), new(need[1], (P, 3))
        g_op, g_sc = new(need[2], tuple(ctx.opacity_shape)), new(need[3], (P, 3))
        g_rot, g_raw = new(need[4], (P, 4)), new(need[5], (P, 4))
        # The SH rows: shs = features (+ d_sh on the dynamic rows), so with both regions rendered the features' gradient IS
        # the incoming one, row for row -- it is handed on as it is (no [P, M, 3] tensor allocated, no 320 bytes per
        # Gaussian read and written again; the kernel then only gathers the dynamic rows for d_sh).  With a region left
        # out its rows must come out zero whatever arrives: the copy stays.
        alias = bool(rs and rd) and _ALIAS_SH_GRADS
        alias_fc = alias and need[6] and g_shs is not None and g_shs.dtype == torch.float32 and g_shs.is_contiguous()
        alias_fp = alias and need[7] and g_shs_p is not None and g_shs_p.dtype == torch.float32 and g_shs_p.is_contiguous()
        g_fc = None if alias_fc else new(need[6], (P, M, 3))
        g_fp = None if alias_fp else new(need[7], (P, M_p, 2))
        nd = ctx.off_rows
        g_dxyz = new(need[9] and nd[0] >= 0 and ctx.offs_scalar[0] is None, (nd[0], 3))
        g_drot = new(need[10] and ctx.offs_scalar[1] is None, (nd[1], 4))
        g_dsh = new(need[11] and ctx.offs_scalar[2] is None, (nd[2], M, 3))
        g_dshp = new(need[12] and ctx.offs_scalar[3] is None, (nd[3], M_p, 2))

        gc = lambda t, n: None if t is None else _f32c(t, dev, "grad_" + n)
        gs = [gc(t, n) for t, n in ((g_means3D, "means3D"), (g_means2D, "means2D"), (g_opacity, "opacity"),
                                    (g_scales, "scales"), (g_rotations, "rotations"), (g_shs, "shs"), (g_shs_p, "shs_p"))]
        io = _lib.AssembleBwdIO()
        io.scratch, io.rotation_raw = scratch.data_ptr(), _p(raw_c)
        if ctx.offs_scalar[1] is None:
            io.d_rot = _p(d_rot_c)
        else:
            io.d_rot_scalar = ctx.offs_scalar[1]
        (io.g_means3D, io.g_means2D, io.g_opacity, io.g_scales, io.g_rotations, io.g_shs, io.g_shs_p) = [_p(t) for t in gs]
        io.g_xyz, io.g_screenspace, io.g_opacity_in, io.g_scaling = _p(g_xyz), _p(g_ssp), _p(g_op), _p(g_sc)
        io.g_rotation, io.g_rotation_raw, io.g_feat_color, io.g_feat_phasor = _p(g_rot), _p(g_raw), _p(g_fc), _p(g_fp)
        io.g_d_xyz, io.g_d_rot, io.g_d_sh, io.g_d_sh_p = _p(g_dxyz), _p(g_drot), _p(g_dsh), _p(g_dshp)
        io.static_from_raw = int(ctx.static_from_raw)
        stream = _lib.raw_stream(dev)
        with _lib.on_device(dev):
            _lib.check(lib.gft_assemble_backward(stream, P, M, M_p, rs, rd, C.byref(io)))
        if alias_fc:
            g_fc = g_shs if tuple(g_shs.shape) == (P, M, 3) else g_shs.view(P, M, 3)
        if alias_fp:
            g_fp = g_shs_p if tuple(g_shs_p.shape) == (P, M_p, 2) else g_shs_p.view(P, M_p, 2)
        return (g_xyz, g_ssp, g_op, g_sc, g_rot, g_raw, g_fc, g_fp, None, g_dxyz, g_drot, g_dsh, g_dshp,
                None, None, None)


def assemble_inputs(xyz, screenspace_points, opacity, scaling, rotation, rotation_raw, features_color,
                    features_phasor, motion_mask, d_xyz=0.0, d_rot=0.0, d_sh=0.0, d_sh_p=0.0,
                    render_regions=("static", "dynamic"), validate=False):
    """Returns ``(means3D, means2D, opacity, scales, rotations, shs, shs_p)`` exactly as lines 81-105
    of the reference's ``gaussian_renderer/__init__.py`` build them.  The ``d_*`` tensors must share one row
    count (checked on the host); a dynamic Gaussian beyond their last row gets NaN outputs and zero gradients
    (never an out-of-bounds access).  ``validate=True`` adds the reference's exact row-count check against the
    number of True entries of the mask (one blocking read) and raises like the reference's masked assignment.

    ``rotation=None``: the static rows' rotations are ``normalize(rotation_raw)`` computed inside the kernels (what
    ``pc.get_rotation`` is), forward and backward -- the caller's eager normalize and its eight backward launches over
    [P, 4] drop out; the gradient of those rows then arrives in ``rotation_raw.grad``."""
    return _AssembleInputs.apply(xyz, screenspace_points, opacity, scaling, rotation, rotation_raw,
                                 features_color, features_phasor, motion_mask, d_xyz, d_rot, d_sh, d_sh_p,
                                 "static" in render_regions, "dynamic" in render_regions, validate)


class _AssembleParameters(torch.autograd.Function):
    """assemble_inputs over the tensors the model keeps: the activations and concatenations of ``pc.get_*`` are done in the
    assembly's kernels, forward and backward (include/gftorf_assemble.h: opacity_is_raw, scaling_is_raw, feat_dc_color ...)."""

    @staticmethod
    def forward(ctx, xyz, screenspace, opacity_raw, scaling_raw, rotation_raw, f_dc, f_rest, phase_dc, phase_rest, amp_dc, amp_rest,
                motion_mask, d_xyz, d_rot, d_sh, d_sh_p, render_static, render_dynamic):
        lib = _lib.load()
        dev = xyz.device
        if dev.type != "cuda":
            raise RuntimeError("gftorf_amd.assemble_parameters runs on a HIP device only (xyz is on %s); there is no CPU path" % (dev,))
        P = xyz.size(0)
        M, M_p = f_dc.size(1) + f_rest.size(1), phase_dc.size(1) + phase_rest.size(1)
        if f_dc.size(1) != 1 or phase_dc.size(1) != 1 or amp_dc.size(1) != 1 or amp_dc.size(1) + amp_rest.size(1) != M_p:
            raise RuntimeError("assemble_parameters: the dc tensors hold one coefficient, phase and amplitude the same number")
        names = ("xyz", "screenspace_points", "_opacity", "_scaling", "_rotation", "_features_dc_color", "_features_rest_color",
                 "phase_f_dc", "phase_f_rest", "amp_f_dc", "amp_f_rest")
        src = [_f32c(t, dev, n) for t, n in zip((xyz, screenspace, opacity_raw, scaling_raw, rotation_raw, f_dc, f_rest, phase_dc,
                                                 phase_rest, amp_dc, amp_rest), names)]
        xyz_c, ssp_c, op_c, sc_c, raw_c, fdc_c, frest_c, pdc_c, prest_c, adc_c, arest_c = src
        if motion_mask.dtype != torch.bool or motion_mask.numel() != P:
            raise RuntimeError("motion_mask must be a bool tensor with one entry per Gaussian")
        mask_c = motion_mask.to(dev).contiguous()
        offs = []
        for t, n, shape in ((d_xyz, "d_xyz", (3,)), (d_rot, "d_rot", (4,)), (d_sh, "d_sh", (M, 3)), (d_sh_p, "d_sh_p", (M_p, 2))):
            if _is_tensor(t):
                if tuple(t.shape[1:]) != shape:
                    raise RuntimeError("%s has shape %s, expected (num_dynamic, %s)" % (n, tuple(t.shape), ", ".join(map(str, shape))))
                offs.append(_f32c(t, dev, n))
            else:
                offs.append(float(t))
        rows = {n: v.size(0) for v, n in zip(offs, ("d_xyz", "d_rot", "d_sh", "d_sh_p")) if _is_tensor(v)}
        if len(set(rows.values())) > 1:
            raise RuntimeError("shape mismatch: the offset tensors disagree on the number of dynamic Gaussians: %s" % (rows,))
        n_off = next(iter(rows.values())) if rows else 0
        if n_off > P:
            raise RuntimeError("shape mismatch: the offset tensors have %d rows for %d Gaussians" % (n_off, P))
        f32 = dict(device=dev, dtype=torch.float32)
        outs = [torch.empty(s, **f32) for s in ((P, 3), (P, 3), tuple(opacity_raw.shape), (P, 3), (P, 4), (P, M, 3), (P, M_p, 2))]
        scratch = torch.empty((lib.gft_assemble_scratch_bytes(P),), device=dev, dtype=torch.uint8)
        io = _lib.AssembleIO()
        io.xyz, io.screenspace, io.opacity, io.scaling, io.rotation_raw = _p(xyz_c), _p(ssp_c), _p(op_c), _p(sc_c), _p(raw_c)
        io.opacity_is_raw = io.scaling_is_raw = 1
        io.feat_dc_color, io.feat_rest_color = _p(fdc_c), _p(frest_c)
        io.phase_dc, io.phase_rest, io.amp_dc, io.amp_rest = _p(pdc_c), _p(prest_c), _p(adc_c), _p(arest_c)
        io.motion_mask = _p(mask_c)
        for name, v in zip(("d_xyz", "d_rot", "d_sh", "d_sh_p"), offs):
            if _is_tensor(v):
                setattr(io, name, _p(v))
            else:
                setattr(io, name + "_scalar", v)
        io.num_offset_rows = n_off
        io.scratch = scratch.data_ptr()
        (io.out_means3D, io.out_means2D, io.out_opacity, io.out_scales, io.out_rotations, io.out_shs, io.out_shs_p) = [_p(t) for t in outs]
        with _lib.on_device(dev):
            _lib.check(lib.gft_assemble_forward(_lib.raw_stream(dev), P, M, M_p, int(render_static), int(render_dynamic), C.byref(io)))
        ctx.sizes = (P, M, M_p, int(render_static), int(render_dynamic))
        ctx.offs_scalar = [None if _is_tensor(v) else v for v in offs]
        ctx.off_rows = [v.size(0) if _is_tensor(v) else 0 for v in offs]
        ctx.shapes = [tuple(t.shape) for t in (opacity_raw, f_dc, f_rest, phase_dc, phase_rest, amp_dc, amp_rest)]
        ctx.save_for_backward(raw_c, offs[1] if _is_tensor(offs[1]) else raw_c.new_empty(0), scratch, op_c, sc_c)
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_means3D, g_means2D, g_opacity, g_scales, g_rotations, g_shs, g_shs_p):
        lib = _lib.load()
        raw_c, d_rot_c, scratch, op_c, sc_c = ctx.saved_tensors
        P, M, M_p, rs, rd = ctx.sizes
        dev = raw_c.device
        need = ctx.needs_input_grad
        f32 = dict(device=dev, dtype=torch.float32)
        new = lambda want, shape: torch.empty(shape, **f32) if want else None
        op_shape, fdc_s, frest_s, pdc_s, prest_s, adc_s, arest_s = ctx.shapes
        g_xyz, g_ssp = new(need[0], (P, 3)), new(need[1], (P, 3))
        g_op, g_sc, g_raw = new(need[2], op_shape), new(need[3], (P, 3)), new(need[4], (P, 4))
        g_fdc, g_frest = new(need[5], fdc_s), new(need[6], frest_s)
        g_pdc, g_prest, g_adc, g_arest = new(need[7], pdc_s), new(need[8], prest_s), new(need[9], adc_s), new(need[10], arest_s)
        nd = ctx.off_rows
        g_dxyz = new(need[12] and ctx.offs_scalar[0] is None, (nd[0], 3))
        g_drot = new(need[13] and ctx.offs_scalar[1] is None, (nd[1], 4))
        g_dsh = new(need[14] and ctx.offs_scalar[2] is None, (nd[2], M, 3))
        g_dshp = new(need[15] and ctx.offs_scalar[3] is None, (nd[3], M_p, 2))
        gc = lambda t, n: None if t is None else _f32c(t, dev, "grad_" + n)
        gs = [gc(t, n) for t, n in ((g_means3D, "means3D"), (g_means2D, "means2D"), (g_opacity, "opacity"),
                                    (g_scales, "scales"), (g_rotations, "rotations"), (g_shs, "shs"), (g_shs_p, "shs_p"))]
        io = _lib.AssembleBwdIO()
        io.scratch, io.rotation_raw = scratch.data_ptr(), _p(raw_c)
        if ctx.offs_scalar[1] is None:
            io.d_rot = _p(d_rot_c)
        else:
            io.d_rot_scalar = ctx.offs_scalar[1]
        (io.g_means3D, io.g_means2D, io.g_opacity, io.g_scales, io.g_rotations, io.g_shs, io.g_shs_p) = [_p(t) for t in gs]
        io.g_xyz, io.g_screenspace, io.g_opacity_in, io.g_scaling, io.g_rotation_raw = _p(g_xyz), _p(g_ssp), _p(g_op), _p(g_sc), _p(g_raw)
        io.opacity_raw, io.scaling_raw = _p(op_c), _p(sc_c)
        io.g_feat_dc_color, io.g_feat_rest_color = _p(g_fdc), _p(g_frest)
        io.g_phase_dc, io.g_phase_rest, io.g_amp_dc, io.g_amp_rest = _p(g_pdc), _p(g_prest), _p(g_adc), _p(g_arest)
        io.g_d_xyz, io.g_d_rot, io.g_d_sh, io.g_d_sh_p = _p(g_dxyz), _p(g_drot), _p(g_dsh), _p(g_dshp)
        io.static_from_raw = 1
        with _lib.on_device(dev):
            _lib.check(lib.gft_assemble_backward(_lib.raw_stream(dev), P, M, M_p, rs, rd, C.byref(io)))
        return (g_xyz, g_ssp, g_op, g_sc, g_raw, g_fdc, g_frest, g_pdc, g_prest, g_adc, g_arest, None, g_dxyz, g_drot, g_dsh, g_dshp,
                None, None)


def assemble_parameters(xyz, screenspace_points, opacity_raw, scaling_raw, rotation_raw, features_dc_color, features_rest_color,
                        phase_f_dc, phase_f_rest, amp_f_dc, amp_f_rest, motion_mask, d_xyz=0.0, d_rot=0.0, d_sh=0.0, d_sh_p=0.0,
                        render_regions=("static", "dynamic")):
    """``assemble_inputs`` fed with the tensors the model keeps instead of the activated ones -- what ``pc.get_opacity``
    (sigmoid), ``pc.get_scaling`` (exp), ``pc.get_rotation`` (normalize), ``pc.get_features_color`` (cat of dc and rest) and
    ``pc.get_features_phasor`` (cat of cat) compute (scene/gaussian_model.py:123-153) is done inside the assembly's kernels,
    forward and backward: the same seven outputs, gradients in the raw tensors' ``.grad``; about twenty-five eager launches per
    iteration and the activated copies they leave drop out.

        means3D, means2D, opacity, scales, rotations, shs, shs_p = assemble_parameters(
            pc._xyz, screenspace_points, pc._opacity, pc._scaling, pc._rotation, pc._features_dc_color, pc._features_rest_color,
            pc.phase_f_dc, pc.phase_f_rest, pc.amp_f_dc, pc.amp_f_rest, pc.get_motion_mask, d_xyz, d_rot, d_sh, d_sh_p, render_regions)
    """
    return _AssembleParameters.apply(xyz, screenspace_points, opacity_raw, scaling_raw, rotation_raw, features_dc_color,
                                     features_rest_color, phase_f_dc, phase_f_rest, amp_f_dc, amp_f_rest, motion_mask, d_xyz, d_rot,
                                     d_sh, d_sh_p, "static" in render_regions, "dynamic" in render_regions)
