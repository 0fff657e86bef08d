"""Opt-in: the two rasterizer calls of one training iteration as one autograd node.

The reference's ``render()`` calls the rasterizer twice on the same Gaussians, once per camera
(``gaussian_renderer/__init__.py:107-128``: colour camera, then ToF camera with ``phase_offset`` / ``dc_offset``),
and autograd adds the two sets of dense per-Gaussian gradients (376 B per Gaussian each).  ``GaussianRasterizerPair``
renders both views in one call -- the reference API is untouched, like ``assemble_inputs`` this is an addition:

* forward: view A is queued on torch's current stream, view B on a side stream, so B's preprocess / binning kernels
  (HBM- and latency-bound) run beside A's render kernel (bound by VALU issue and by its dependency chains; one view
  fills 4800 of the chip's 8192 wave slots) and the two render kernels share the chip.  Each view's outputs are
  bit-identical to a single ``GaussianRasterizer`` call: same kernels, same arguments, other stream.
* backward: ONE set of gradient tensors.  View A's backward writes it (zeros for the Gaussians nobody blended), view
  B's backward then ADDS the rows of the Gaussians it blended (``gft_config.grads_accumulate``) -- no second 376 B per
  Gaussian of zeros and no elementwise sum of two dense sets by autograd (3 x 376 B per Gaussian of traffic).

``means2D`` gets the sum of both views' screen-space gradients, as when the reference passes one tensor to both
calls (``gaussian_renderer/__init__.py:29-34``).
"""
import torch
import torch.nn as nn

from .api import (_scalar, native_forward, native_backward, run_backward)

_side_streams = {}


def _side_stream(dev):
    s = _side_streams.get(dev.index)
    if s is None:
        s = _side_streams[dev.index] = torch.cuda.Stream(device=dev)
    return s


def _absent():
    return torch.Tensor([])


class _RasterizePair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, sh_p, colors_precomp, phasors_precomp, opacities, scales, rotations,
                cov3Ds_precomp, phase_offset_a, dc_offset_a, phase_offset_b, dc_offset_b, settings_a, settings_b):
        offs = [(_scalar(phase_offset_a), _scalar(dc_offset_a)), (_scalar(phase_offset_b), _scalar(dc_offset_b))]
        want_bw = any(ctx.needs_input_grad)
        dev = means3D.device
        overlap = dev.type == "cuda" and means3D.size(0) > 0
        if overlap:
            main = torch.cuda.current_stream(dev)
            side = _side_stream(dev)

            def fork():
                # B's kernels start behind everything queued on the main stream so far: the producers of its inputs AND
                # the contiguous / aligned copies native_forward has just made of them on the main stream (the
                # reference's camera matrices are `.transpose(0, 1)` views, copied on every call)
                ev = torch.cuda.Event()
                ev.record(main)
                side.wait_event(ev)
        ra = native_forward(settings_a, means3D, sh, sh_p, colors_precomp, phasors_precomp, opacities, scales, rotations,
                            cov3Ds_precomp, offs[0][0], offs[0][1], want_bw, True, hint_slot=1)
        # (B's buffers come from the current stream's pool like A's: it is joined below before anything is returned,
        # and A's call frees nothing that B could be handed while A's kernels still use it)
        rb = native_forward(settings_b, means3D, sh, sh_p, colors_precomp, phasors_precomp, opacities, scales, rotations,
                            cov3Ds_precomp, offs[1][0], offs[1][1], want_bw, True,
                            stream=side.cuda_stream if overlap else None, hint_slot=2, share_grads=ra["prep"],
                            pre_launch=fork if overlap else None,
                            acc_any_stream=overlap)      # (a kept accumulator's last kernels ran on the main stream: fork() orders them)
        if overlap:
            join = torch.cuda.Event()
            join.record(side)
            main.wait_event(join)
        ctx.settings = (settings_a, settings_b)
        ctx.offs = offs
        ctx.want_bw = want_bw
        ctx.preps = [ra["prep"], rb["prep"]]
        ctx.extra = [(r["bg"], r["consts"]) for r in (ra, rb)]
        means3D_c, opac_c, sh_c, sh_p_c, scales_c, rot_c, cov_c, colors_c, phasors_c = ra["inputs"]
        ctx.present = (sh_c is not None, sh_p_c is not None, colors_c is not None, phasors_c is not None,
                       scales_c is not None, cov_c is not None)
        shape = lambda x: x.shape if isinstance(x, torch.Tensor) else None
        ctx.in_shapes = (opacities.shape, shape(phase_offset_a), shape(dc_offset_a), shape(phase_offset_b), shape(dc_offset_b))
        ctx.set_materialize_grads(False)
        dummy = means3D_c.new_empty(0)
        opt = lambda t: t if t is not None else dummy
        ctx.save_for_backward(means3D_c, opt(opac_c), opt(sh_c), opt(sh_p_c), opt(scales_c), opt(rot_c), opt(cov_c),
                              ra["outputs"][10], ra["geom"], ra["binning"], ra["img"],
                              rb["outputs"][10], rb["geom"], rb["binning"], rb["img"],
                              ra["outputs"][8], rb["outputs"][8])       # `pixels` of both views (read by the backward)
        ctx.mark_non_differentiable(ra["outputs"][10], rb["outputs"][10])
        return tuple(ra["outputs"]) + tuple(rb["outputs"])

    @staticmethod
    def backward(ctx, *grads):
        (means3D, opac, sh, sh_p, scales, rotations, cov3D, radii_a, geom_a, bin_a, img_a,
         radii_b, geom_b, bin_b, img_b, _pix_a, _pix_b) = ctx.saved_tensors
        has_sh, has_sh_p, has_colors, has_phasors, has_scales, has_cov = ctx.present
        views = [(ctx.settings[0], radii_a, geom_a, bin_a, img_a, grads[0:11]),
                 (ctx.settings[1], radii_b, geom_b, bin_b, img_b, grads[11:22])]
        preps, ctx.preps = ctx.preps, [None, None]
        g, offsets = None, [None, None]
        for v, (s, radii, geom, binning, img, gv) in enumerate(views):
            # color, phasor, depth, acc, depth_distortion; the other planes' gradients are accepted and ignored
            go = (gv[0], gv[1], gv[2], gv[4], gv[6])
            if all(t is None for t in go):
                continue                              # this view reached no loss (reference: its backward never runs)
            prep = preps[v]
            if prep is not None:
                # the second view adds to the first one's tensors -- unless it is the only one with a gradient
                prep["cfg"].grads_accumulate = int(g is not None)
                res = run_backward(prep, go, geom, binning, img, bool(s.debug))
            else:                                     # second backward through the same forward (retain_graph)
                res = native_backward(s, means3D, opac, sh if has_sh else None, sh_p if has_sh_p else None,
                                      scales if has_scales else None, rotations if has_scales else None,
                                      cov3D if has_cov else None, radii, geom, binning, img, ctx.extra[v][0], ctx.extra[v][1],
                                      ctx.offs[v][0], ctx.offs[v][1], go, None, has_colors, has_cov, ctx.want_bw)
                if g is not None:
                    for k in g:
                        if k != "offsets" and g[k] is not None:
                            g[k] = g[k] + res[k]
                    res = dict(g, offsets=res["offsets"])
            offsets[v] = res["offsets"]
            g = res
        if g is None:
            return (None,) * 16
        op_shape, pa, da, pb, db = ctx.in_shapes
        sa, sb = ctx.settings

        def off(view, settings_flag, shape, i):
            if not settings_flag or shape is None or offsets[view] is None:
                return None
            return offsets[view][i:i + 1].reshape(shape)
        return (g["means3D"], g["means2D"], g["sh"], g["sh_p"], g["colors"], None,
                g["opacities"].reshape(op_shape), g["scales"], g["rotations"], g["cov3D"],
                off(0, sa.optimize_phase_offset, pa, 0), off(0, sa.optimize_dc_offset, da, 1),
                off(1, sb.optimize_phase_offset, pb, 0), off(1, sb.optimize_dc_offset, db, 1), None, None)


class GaussianRasterizerPair(nn.Module):
    """``GaussianRasterizerPair(settings_a, settings_b)(means3D=..., ...)`` = the pair
    ``(GaussianRasterizer(settings_a)(...), GaussianRasterizer(settings_b)(...))`` on the same Gaussians: two 11-tuples
    in the reference's order.  ``phase_offset`` / ``dc_offset`` take one value per view (a float or 1-element tensor
    each, as in ``gaussian_renderer/__init__.py:126-127``; the reference's colour-camera call passes none: 0.0)."""

    def __init__(self, raster_settings_a, raster_settings_b):
        super().__init__()
        self.raster_settings = (raster_settings_a, raster_settings_b)

    def forward(self, means3D, means2D, opacities, shs=None, shs_p=None, colors_precomp=None, phasors_precomp=None,
                scales=None, rotations=None, cov3D_precomp=None, phase_offset=(0.0, 0.0), dc_offset=(0.0, 0.0)):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        a = lambda t: _absent() if t is None else t
        sa, sb = self.raster_settings
        out = _RasterizePair.apply(means3D, means2D, a(shs), a(shs_p), a(colors_precomp), a(phasors_precomp), opacities,
                                   a(scales), a(rotations), a(cov3D_precomp), phase_offset[0], dc_offset[0],
                                   phase_offset[1], dc_offset[1], sa, sb)
        return out[:11], out[11:]


def render_pair(raster_settings_a, raster_settings_b, **kwargs):
    """Functional form of :class:`GaussianRasterizerPair`."""
    return GaussianRasterizerPair(raster_settings_a, raster_settings_b)(**kwargs)
