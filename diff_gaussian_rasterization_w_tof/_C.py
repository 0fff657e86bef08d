"""``diff_gaussian_rasterization_w_tof._C`` -- the pybind-level entry points of the reference's extension
(``RAST/ext.cpp:15-19``) on MI355X, same names, positional argument order and return tuples as
``RAST/rasterize_points.h:18-88``:

  * ``rasterize_gaussians(27 args)``          -> ``(num_rendered, color, phasor, depth, normal, acc, entropy,
    depth_distortion, amp_distortion, pixels, distribution, radii, geomBuffer, binningBuffer, imgBuffer)``
    (``RasterizeGaussiansCUDA``, rasterize_points.cu:42-165)
  * ``rasterize_gaussians_backward(36 args)`` -> ``(dL_dmeans2D, dL_dcolors, dL_dphasors, dL_dopacity, dL_dmeans3D,
    dL_dcov3D, dL_dsh, dL_dsh_p, dL_dscales, dL_drotations, dL_dphase_offset, dL_ddc_offset)``
    (``RasterizeGaussiansBackwardCUDA``, rasterize_points.cu:167-281)
  * ``mark_visible(means3D, viewmatrix, projmatrix, znear, zfar)`` -> ``bool[P]`` (rasterize_points.cu:283-304)

This is INTEGRATION.md's Route B: the reference's own Python wrapper (``_RasterizeGaussians`` of
``RAST/diff_gaussian_rasterization_w_tof/__init__.py:69-206``) runs unchanged on top of this module.  The work is
done by libgftorf_rast.so through the C ABI of ``include/gftorf_rast.h``; there is no CPU path.

Differences from the reference's binary interface, all of them in slots its wrapper never reads:

  * the three buffers are this library's scratch layouts (``gft_get_layout``), opaque byte tensors as in the
    reference; the binning buffer may hold more than ``num_rendered`` instances (it is sized from the previous
    frame so that the forward needs no host round trip) -- its capacity is read back from its size;
  * ``dL_dphasors`` (slot 2 of the backward's tuple) is ``None``: the reference returns its internal ``[P,7]``
    per-Gaussian phasor gradient there, which its wrapper hands to autograd as the gradient of the ``[P,2]``
    ``phasors_precomp`` input (never differentiable, ``backward.cu:527``); this library accumulates the 7 planes on
    their rank-3 basis and never forms that array;
  * ``dL_dcolors`` / ``dL_dcov3D`` are always returned (like the reference); the gradients of absent inputs
    (``sh``, ``sh_p``, ``scales`` / ``rotations``) are empty tensors where the reference returns zero-filled ones
    of zero rows' worth of coefficients (``M = 0``), i.e. the same shapes.
"""
import torch

from gftorf_amd import api as _api


def _empty(t):
    return t is None or (isinstance(t, torch.Tensor) and t.numel() == 0)


def rasterize_gaussians(background, means3D, colors, phasors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                        viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, sh_p, degree, campos,
                        prefiltered, debug, near_n, far_n, depth_range, use_view_dependent_phase, phase_offset, dc_offset):
    s = _api._Settings(bg=background, scale_modifier=scale_modifier, viewmatrix=viewmatrix, projmatrix=projmatrix,
                       tanfovx=tan_fovx, tanfovy=tan_fovy, image_height=image_height, image_width=image_width,
                       sh_degree=degree, campos=campos, prefiltered=prefiltered, debug=debug, near_n=near_n, far_n=far_n,
                       depth_range=depth_range, use_view_dependent_phase=use_view_dependent_phase)
    # whether a backward follows is not known at this level: the forward always leaves what it needs
    r = _api.native_forward(s, means3D, sh, sh_p, colors, phasors, opacity, scales, rotations, cov3D_precomp,
                            _api._scalar(phase_offset), _api._scalar(dc_offset), True, False)
    return (r["R"],) + tuple(r["outputs"]) + (r["geom"], r["binning"], r["img"])


def rasterize_gaussians_backward(background, means3D, radii, colors, phasors, scales, rotations, scale_modifier,
                                 cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, dL_dout_phasor,
                                 dL_dout_depth, dL_dout_normal, dL_dout_acc, dL_dout_entropy, dL_dout_depth_distortion,
                                 dL_dout_amp_distortion, sh, sh_p, degree, campos, geomBuffer, R, binningBuffer,
                                 imageBuffer, debug, near_n, far_n, depth_range, use_view_dependent_phase, phase_offset,
                                 dc_offset):
    # H, W come from the upstream colour gradient as in the reference (rasterize_points.cu:201-202)
    H, W = int(dL_dout_color.size(1)), int(dL_dout_color.size(2))
    s = _api._Settings(bg=background, scale_modifier=scale_modifier, viewmatrix=viewmatrix, projmatrix=projmatrix,
                       tanfovx=tan_fovx, tanfovy=tan_fovy, image_height=H, image_width=W, sh_degree=degree, campos=campos,
                       prefiltered=False, debug=debug, near_n=near_n, far_n=far_n, depth_range=depth_range,
                       use_view_dependent_phase=use_view_dependent_phase)
    dev = means3D.device
    P = means3D.size(0)
    c = lambda t, n: None if _empty(t) else _api._f32(t, dev, n)
    means3D_c = _api._f32(means3D, dev, "means3D") if P else means3D
    sh_c, sh_p_c = c(sh, "shs"), c(sh_p, "shs_p")
    scales_c, rot_c, cov_c = c(scales, "scales"), c(rotations, "rotations"), c(cov3D_precomp, "cov3D_precomp")
    bg = _api._bg_strides(background, H, W, dev)
    consts = (_api._f32(viewmatrix, dev, "viewmatrix"), _api._f32(projmatrix, dev, "projmatrix"),
              _api._f32(campos, dev, "campos"))
    # no opacity argument at this level (rasterize_points.h:55-88): the kernels read the value the forward stored
    g = _api.native_backward(s, means3D_c, None, sh_c, sh_p_c, scales_c, rot_c, cov_c, radii, geomBuffer,
                             binningBuffer, imageBuffer, bg, consts, _api._scalar(phase_offset),
                             _api._scalar(dc_offset),
                             (dL_dout_color, dL_dout_phasor, dL_dout_depth, dL_dout_acc, dL_dout_depth_distortion),
                             None, True, True, True)
    f32 = dict(device=dev, dtype=torch.float32)
    z = lambda *shape: torch.zeros(shape, **f32)
    dL_dsh = g["sh"] if g["sh"] is not None else z(P, 0, 3)
    dL_dsh_p = g["sh_p"] if g["sh_p"] is not None else z(P, 0, 2)
    dL_dscales = g["scales"] if g["scales"] is not None else z(P, 3)
    dL_drot = g["rotations"] if g["rotations"] is not None else z(P, 4)
    off = g["offsets"]
    return (g["means2D"], g["colors"], None, g["opacities"], g["means3D"], g["cov3D"], dL_dsh, dL_dsh_p, dL_dscales,
            dL_drot, off[0:1], off[1:2])


def mark_visible(means3D, viewmatrix, projmatrix, znear, zfar):
    s = _api._Settings(bg=None, scale_modifier=1.0, viewmatrix=viewmatrix, projmatrix=projmatrix, tanfovx=1.0, tanfovy=1.0,
                       image_height=0, image_width=0, sh_degree=0, campos=None, prefiltered=False, debug=False,
                       near_n=znear, far_n=zfar, depth_range=0.0, use_view_dependent_phase=False)
    return _api.GaussianRasterizer(s).markVisible(means3D)
