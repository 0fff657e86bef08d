"""Shared helpers of the parity tests: build a scene, run it through the CPU
oracle and through the HIP path (public Python API -> C ABI), compare."""
import numpy as np
import torch

from gftorf_amd import synth

GRAD_KEYS = ["color", "phasor", "depth", "acc", "depth_distortion"]


def small_scene(P=400, W=80, H=48, seed=3, D=3, sh_coeffs=16, scale_lo=0.01, scale_hi=0.12,
                w2c="tilted", spread=1.05, tof=True, opacity=None, z_lo=1.0, z_hi=5.5):
    if isinstance(w2c, str):
        w2c = synth.look_at_w2c(0.15, -0.1, 0.05, (0.1, -0.05, 0.2)) if w2c == "tilted" else None
    cam = synth.make_camera(W, H, w2c=w2c)
    g = synth.make_gaussians(P, cam, seed, sh_coeffs=sh_coeffs, scale_lo=scale_lo, scale_hi=scale_hi,
                             spread=spread, z_lo=z_lo, z_hi=z_hi)
    if not tof:
        g["shs_p"] = None
    if opacity is not None:
        g["opacities"] = np.full_like(g["opacities"], opacity)
    return dict(cfg=dict(P=P, W=W, H=H, D=D, sh_coeffs=sh_coeffs, tof=tof), cam=cam, gaussians=g,
                bg=synth.make_background(W, H, seed), grads=synth.make_pixel_grads(W, H, seed),
                depth_range=10.0, phase_offset=0.1, dc_offset=0.05, use_view_dependent_phase=True)


def oracle_kwargs(scene, **over):
    cam, cfg = scene["cam"], scene["cfg"]
    kw = dict(bg=scene["bg"], viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"],
              campos=cam["campos"], image_height=cfg["H"], image_width=cfg["W"],
              tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=cfg["D"],
              near_n=cam["znear"], far_n=cam["zfar"], depth_range=scene["depth_range"],
              use_view_dependent_phase=scene["use_view_dependent_phase"],
              phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"])
    kw.update(over)
    return kw


def run_oracle(oracle, scene, backward=True, inputs=None, **over):
    g = dict(scene["gaussians"])
    if inputs:
        g.update(inputs)
    kw = oracle_kwargs(scene, **over)
    f = oracle.forward(g["means3D"], g["opacities"], shs=g.get("shs"), shs_p=g.get("shs_p"),
                       colors_precomp=g.get("colors_precomp"), phasors_precomp=g.get("phasors_precomp"),
                       scales=g.get("scales"), rotations=g.get("rotations"),
                       cov3D_precomp=g.get("cov3D_precomp"), **kw)
    b = None
    if backward:
        gr = scene["grads"]
        b = oracle.backward(f, gr["color"], gr["phasor"], gr["depth"], gr["acc"], gr["depth_distortion"])
    return f, b


def gpu_settings(scene, dev, bg=None, debug=False, optimize_offsets=False, **over):
    from gftorf_amd import GaussianRasterizationSettings
    cam, cfg = scene["cam"], scene["cfg"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    kw = dict(image_height=cfg["H"], image_width=cfg["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
              bg=t(scene["bg"]) if bg is None else bg, scale_modifier=1.0,
              viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=cfg["D"],
              campos=t(cam["campos"]), prefiltered=False, debug=debug, near_n=cam["znear"],
              far_n=cam["zfar"], depth_range=scene["depth_range"],
              use_view_dependent_phase=scene["use_view_dependent_phase"],
              optimize_phase_offset=optimize_offsets, optimize_dc_offset=optimize_offsets)
    kw.update(over)
    return GaussianRasterizationSettings(**kw)


OUT_NAMES = ["color", "phasor", "depth", "normal", "acc", "entropy", "depth_distortion",
             "amp_distortion", "pixels", "distribution", "radii"]


def run_gpu(scene, dev, backward=True, inputs=None, settings=None, optimize_offsets=False, **over):
    """Runs the public API on the HIP device.  Returns (outputs dict of numpy, grads dict of numpy,
    tensors dict)."""
    from gftorf_amd import GaussianRasterizer
    g = dict(scene["gaussians"])
    if inputs:
        g.update(inputs)
    P = g["means3D"].shape[0]
    leaf = {}
    for k, v in g.items():
        if v is None:
            continue
        leaf[k] = torch.tensor(v, dtype=torch.float32, device=dev, requires_grad=backward)
    means2D = torch.zeros((P, 3), dtype=torch.float32, device=dev, requires_grad=backward)
    if settings is None:
        settings = gpu_settings(scene, dev, optimize_offsets=optimize_offsets, **over)
    if optimize_offsets:
        ph = torch.tensor([scene["phase_offset"]], dtype=torch.float32, device=dev, requires_grad=True)
        dc = torch.tensor([scene["dc_offset"]], dtype=torch.float32, device=dev, requires_grad=True)
    else:
        ph, dc = scene["phase_offset"], scene["dc_offset"]
    rast = GaussianRasterizer(raster_settings=settings)
    outs = rast(means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"],
                shs=leaf.get("shs"), shs_p=leaf.get("shs_p"), colors_precomp=leaf.get("colors_precomp"),
                phasors_precomp=leaf.get("phasors_precomp"), scales=leaf.get("scales"),
                rotations=leaf.get("rotations"), cov3D_precomp=leaf.get("cov3D_precomp"),
                phase_offset=ph, dc_offset=dc)
    assert len(outs) == 11
    o = dict(zip(OUT_NAMES, outs))
    grads = None
    if backward:
        gr = scene["grads"]
        loss = sum((o[k] * torch.tensor(gr[k], device=dev)).sum() for k in GRAD_KEYS)
        loss.backward()
        grads = {k: (v.grad.detach().cpu().numpy() if v.grad is not None else None) for k, v in leaf.items()}
        grads["means2D"] = means2D.grad.detach().cpu().numpy()
        if optimize_offsets:
            grads["phase_offset"] = ph.grad.detach().cpu().numpy()
            grads["dc_offset"] = dc.grad.detach().cpu().numpy()
    torch.cuda.synchronize()
    out_np = {k: v.detach().cpu().numpy() for k, v in o.items()}
    return out_np, grads, dict(leaf=leaf, outs=o, means2D=means2D)


def rel_err(ref, got):
    ref = np.asarray(ref, np.float64)
    got = np.asarray(got, np.float64)
    den = np.abs(ref).max()
    return float(np.abs(ref - got).max() / (den + 1e-30)), float(den)


def assert_close(name, ref, got, rtol_max=2e-4, atol=1e-6, frac_bad=0.0, rtol_elem=None):
    """max-norm relative check: |ref-got|_inf <= rtol_max * |ref|_inf + atol.
    frac_bad > 0 tolerates that fraction of elements outside an element-wise band
    (discrete skip/termination flips on borderline alphas)."""
    ref = np.asarray(ref, np.float64)
    got = np.asarray(got, np.float64)
    assert ref.shape == got.shape, "%s: shape %s vs %s" % (name, ref.shape, got.shape)
    if ref.size == 0:
        return
    assert np.isfinite(got).all(), "%s: non-finite values" % name
    den = np.abs(ref).max()
    err = np.abs(ref - got)
    if frac_bad > 0.0:
        tol = (rtol_elem or rtol_max) * den + atol
        bad = (err > tol).mean()
        assert bad <= frac_bad, "%s: %.3g of elements differ by more than %.3g (max err %.3g, scale %.3g)" % (
            name, bad, tol, err.max(), den)
    else:
        assert err.max() <= rtol_max * den + atol, "%s: max err %.3g > %.3g*%.3g+%.3g" % (
            name, err.max(), rtol_max, den, atol)


class _OracleRasterize(torch.autograd.Function):
    """The CPU oracle behind the operator's autograd surface (CPU tensors): forward = oracle.forward, backward =
    oracle.backward.  Stand-in for the HIP rasterizer where a composed step is rehearsed without a GPU."""

    @staticmethod
    def forward(ctx, oracle, kw, means3D, means2D, opacities, shs, shs_p, scales, rotations):
        n = lambda t: t.detach().cpu().numpy()
        f = oracle.forward(n(means3D), n(opacities), shs=n(shs), shs_p=n(shs_p), scales=n(scales), rotations=n(rotations), **kw)
        ctx.oracle, ctx.f, ctx.op_shape = oracle, f, opacities.shape
        t = lambda a: torch.tensor(np.asarray(a))
        outs = (t(f.color), t(f.phasor), t(f.depth), t(f.normal), t(f.acc), t(f.entropy), t(f.depth_distortion),
                t(f.amp_distortion), t(f.pixels), t(f.distribution), t(f.radii))
        ctx.mark_non_differentiable(outs[10])
        return outs

    @staticmethod
    def backward(ctx, g_color, g_phasor, g_depth, _gn, g_acc, _ge, g_dd, *_rest):
        f = ctx.f
        z = lambda g, c: np.zeros((c, f.H, f.W), np.float32) if g is None else g.detach().numpy()
        b = ctx.oracle.backward(f, z(g_color, 3), z(g_phasor, 7), z(g_depth, 1), z(g_acc, 1), z(g_dd, 1))
        t = lambda a: torch.tensor(np.asarray(a, np.float32))
        return (None, None, t(b["dL_dmeans3D"]), t(b["dL_dmeans2D"]), t(b["dL_dopacity"]).reshape(ctx.op_shape), t(b["dL_dsh"]),
                t(b["dL_dsh_p"]), t(b["dL_dscales"]), t(b["dL_drotations"]))


def oracle_rasterizer(oracle, scene, **over):
    """render(frame_id, means3D=, means2D=, opacities=, shs=, shs_p=, scales=, rotations=) -> 11-tuple, on the CPU oracle."""
    kw = oracle_kwargs(scene, **over)

    def render(frame_id, means3D, means2D, opacities, shs, shs_p, scales, rotations):
        return _OracleRasterize.apply(oracle, kw, means3D, means2D, opacities, shs, shs_p, scales, rotations)
    return render
