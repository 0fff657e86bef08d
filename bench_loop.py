"""BASELINE.json config 3 for bench.py: the torf `copier`-shaped optimisation loop on synthetic data.

The real scene is unreachable (no network), so the loop runs on a synthetic one of the reference's sizes
(configs/torf.json:15-23: 100 000 initial Gaussians, 320x240 colour and ToF images, 30 views, SH degree 3, dynamic,
warm_up 2000) and keeps the reference loop's SHAPE, statement for statement (train.py:118-279, 441-474):

    per iteration: LR schedule (gaussian_model.py:294-310) -> SH degree +1 every 1000 its -> random view ->
    random background seeded by the iteration (train.py:120-129) -> after warm_up one DeformNetwork query for the
    frame's time (all Gaussians are dynamic in a torf scene, train.py:110-111) -> parameter activations
    (gaussian_model.py:123-153) -> render(): input assembly + colour-camera and ToF-camera rasterizer calls
    (gaussian_renderer/__init__.py:19-139) -> ToF loss lambda_tof * (0.8 L2 + 0.2 (1 - SSIM)) on the first two phasor
    planes (train.py:209-231; lambda_color is 0 in torf.json) -> backward -> densification statistics
    (train.py:441-449) -> Adam on the Gaussians and, after warm_up, on the network (train.py:467-474).

Densification / pruning / opacity reset, logging, checkpoints and the debug image dumps are left out.
Everything between the statements is this package (assemble_inputs, GaussianRasterizer, DeformNetwork, densify,
FusedAdam); the losses and activations are stock torch, as in the reference.
"""
import math
import random

import numpy as np


def expon_lr(lr_init, lr_final, max_steps, delay_mult=1.0, delay_steps=0):
    """Log-linear interpolation from lr_init to lr_final over max_steps (the schedule utils/general_utils.py:41-77
    describes: exponential decay with an optional eased start)."""
    def f(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        rate = 1.0
        if delay_steps > 0:
            rate = delay_mult + (1 - delay_mult) * math.sin(0.5 * math.pi * min(max(step / delay_steps, 0.0), 1.0))
        t = min(max(step / max(max_steps, 1), 0.0), 1.0)
        return rate * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return f


def make_ssim(channels, dev, size=11, sigma=1.5):
    """Gaussian-window SSIM (utils/loss_utils.py:84-114 computes the standard one) as a closure over its window."""
    import torch
    import torch.nn.functional as F
    g = torch.tensor([math.exp(-(x - size // 2) ** 2 / (2.0 * sigma ** 2)) for x in range(size)])
    g = (g / g.sum()).unsqueeze(1)
    win = (g @ g.t()).float().expand(channels, 1, size, size).contiguous().to(dev)
    C1, C2 = 0.01 ** 2, 0.03 ** 2

    def ssim(a, b):
        a, b = a.unsqueeze(0), b.unsqueeze(0)
        blur = lambda x: F.conv2d(x, win, padding=size // 2, groups=channels)
        mu1, mu2 = blur(a), blur(b)
        s1 = blur(a * a) - mu1 * mu1
        s2 = blur(b * b) - mu2 * mu2
        s12 = blur(a * b) - mu1 * mu2
        return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).mean()
    return ssim


C3 = dict(P=100_000, W=320, H=240, views=30, sh_degree=3, warm_up=2000, depth_range=10.0,
          position_lr_init=0.00016, position_lr_final=0.0000016, position_lr_max_steps=30000, feature_lr=0.0025,
          feature_phase_lr=0.00016, feature_amp_lr=0.00016, opacity_lr=0.05, scaling_lr=0.001, rotation_lr=0.001,
          deform_lr_init=0.0008, deform_lr_final=0.0000016, lambda_tof=1.0, lambda_dssim=0.2, num_phasor_channels=2,
          label="torf 'copier'-shaped optimisation loop (synthetic stand-in): 100k Gaussians, 320x240, 30 views, "
                "deform network on after warm_up 2000, colour + ToF rasterizer call per iteration")


def build_loop(dev, cfg=C3, seed=1236, pair=False, graph=False, fused_loss=True, fused_params=None):
    """Returns (iteration_fn, info): iteration_fn(it) runs iteration `it` (1-based) and returns the loss tensor.

    ``graph=True``: the iteration's device work -- deform query, activations, input assembly, both rasterizer calls, loss,
    backward, densification statistics, both Adam steps -- is captured once per (active SH degree, network on / off) with
    ``torch.cuda.graph`` and replayed; what changes from one iteration to the next reaches the replay through static
    tensors (the drawn view's camera matrices, time and ground truth are copied in, the background is drawn into its
    tensor) and through pinned memory (the learning rates: ``FusedAdam(capturable=True)``).  The same statements in the
    same order on the same values as the eager loop; the first iterations of every configuration run eagerly.

    ``fused_params`` (default: on; ``GFT_LOOP_TORCH_ACTIVATIONS=1`` or ``--torch-loss`` runs: off): the activations and
    concatenations of ``pc.get_*`` inside the assembly (``gftorf_amd.assemble_parameters``) instead of ~25 eager launches.

    ``fused_loss=True`` (default): the two image terms of the loss -- ``l2_loss`` and ``ssim`` of utils/loss_utils.py -- from
    ``gftorf_amd.loss.ssim_l2`` (one launch forward, one backward) instead of eight torch convolutions and ~25 elementwise
    launches; ``False``: stock torch, as the reference has it."""
    import torch
    from gftorf_amd import (FusedAdam, GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizerPair,
                            assemble_inputs, assemble_parameters, densify, reference_network, synth)
    from gftorf_amd import loss as gft_loss
    P, W, H, V = cfg["P"], cfg["W"], cfg["H"], cfg["views"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    if fused_params is None:       # (the stock-torch variant of the loop keeps the reference's eager activations too)
        import os
        fused_params = fused_loss and os.environ.get("GFT_LOOP_TORCH_ACTIVATIONS", "0") != "1"

    # 30 views on a small arc; the ToF sensor sits beside the colour camera (its own pose, cameras.py:121-146)
    base_cam = synth.make_camera(W, H)
    g0 = synth.make_gaussians(P, base_cam, seed, sh_coeffs=16, scale_lo=0.004, scale_hi=0.04)
    cams = []
    for v in range(V):
        a = (v / (V - 1) - 0.5) * 0.30
        w2c_c = synth.look_at_w2c(yaw=a, pitch=0.04 * math.sin(3 * a), t=(-3.2 * math.sin(a), 0.0, 3.2 * (1 - math.cos(a))))
        w2c_t = synth.look_at_w2c(yaw=a, pitch=0.04 * math.sin(3 * a), t=(-3.2 * math.sin(a) + 0.03, 0.0, 3.2 * (1 - math.cos(a))))
        cams.append((synth.make_camera(W, H, w2c=w2c_c), synth.make_camera(W, H, w2c=w2c_t)))

    def settings(cam, bg, degree, tof):
        return GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
            viewmatrix=cam["_view"], projmatrix=cam["_proj"], sh_degree=degree, campos=cam["_campos"], prefiltered=False,
            debug=False, near_n=cam["znear"], far_n=cam["zfar"], depth_range=cfg["depth_range"] if tof else 100.0,
            use_view_dependent_phase=tof)
    for pair in cams:
        for cam in pair:
            cam["_view"], cam["_proj"], cam["_campos"] = t(cam["viewmatrix"]), t(cam["projmatrix"]), t(cam["campos"])

    # ---- ground truth: the same population, displaced along a time-dependent path, rendered once per view
    rng = np.random.default_rng(seed + 1)
    drift = t(rng.normal(0, 0.03, (P, 3)))
    phase_offset = 0.1
    gt = []
    with torch.no_grad():
        zero_bg = torch.zeros((7, H, W), device=dev)
        for v, (cc, ct) in enumerate(cams):
            tv = v / (V - 1)
            xyz_gt = t(g0["means3D"]) + drift * math.sin(2 * math.pi * tv)
            rast = GaussianRasterizer(settings(ct, zero_bg, 3, True))
            out = rast(means3D=xyz_gt, means2D=torch.zeros((P, 3), device=dev), opacities=t(g0["opacities"]), shs=t(g0["shs"]),
                       shs_p=t(g0["shs_p"]), scales=t(g0["scales"]), rotations=t(g0["rotations"]), phase_offset=phase_offset,
                       dc_offset=0.0)
            gt.append(out[1][:cfg["num_phasor_channels"]].clone())

    # ---- the model: GaussianModel's parameters (scene/gaussian_model.py:142-153, 180-236), started away from the truth
    inv_sig = lambda x: np.log(x / (1 - x))
    par = dict(
        xyz=t(g0["means3D"]) + t(rng.normal(0, 0.01, (P, 3))),
        f_dc_color=t(g0["shs"][:, :1, :]), f_rest_color=torch.zeros((P, 15, 3), device=dev),
        phase_f_dc=t(g0["shs_p"][:, :1, :1]), phase_f_rest=torch.zeros((P, 15, 1), device=dev),
        amp_f_dc=t(g0["shs_p"][:, :1, 1:] * 0.8), amp_f_rest=torch.zeros((P, 15, 1), device=dev),
        opacity=t(inv_sig(np.full((P, 1), 0.1))), scaling=t(np.log(g0["scales"] * 1.2)), rotation=t(g0["rotations"]))
    for p_ in par.values():
        p_.requires_grad_(True)
    ext = 1.0   # cameras_extent of the synthetic rig
    lrs = dict(xyz=cfg["position_lr_init"] * ext, f_dc_color=cfg["feature_lr"], f_rest_color=cfg["feature_lr"] / 20.0,
               phase_f_dc=cfg["feature_phase_lr"] * ext, phase_f_rest=cfg["feature_phase_lr"] * ext / 20.0,
               amp_f_dc=cfg["feature_amp_lr"] * ext ** 2, amp_f_rest=cfg["feature_amp_lr"] * ext ** 2 / 20.0,
               opacity=cfg["opacity_lr"], scaling=cfg["scaling_lr"], rotation=cfg["rotation_lr"])
    opt = FusedAdam([{"params": [par[k]], "lr": lrs[k], "name": k} for k in lrs], lr=0.0, eps=1e-15, capturable=graph)
    xyz_lr = expon_lr(cfg["position_lr_init"] * ext, cfg["position_lr_final"] * ext, cfg["position_lr_max_steps"])

    class _Args:      # DeformNetwork.initialize_weights reads two flags (time_utils.py:83-101)
        isotropic_gaussians = False
        xavier_init_dxyz = False
    torch.manual_seed(seed)
    net = reference_network()
    net.initialize_weights(_Args())
    net = net.to(dev)
    opt_net = FusedAdam([{"params": list(net.parameters()), "lr": cfg["deform_lr_init"], "name": "deform"}], lr=0.0, eps=1e-15,
                        capturable=graph)
    net_lr = expon_lr(cfg["deform_lr_init"], cfg["deform_lr_final"], cfg["position_lr_max_steps"] - cfg["warm_up"])

    mask = torch.ones((P,), dtype=torch.bool, device=dev)           # torf: every Gaussian is dynamic
    stats = [torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev), torch.zeros((P,), device=dev)]
    state = dict(degree=0, stack=[], ssim=None, ssim_ok=True)
    try:
        state["ssim"] = make_ssim(cfg["num_phasor_channels"], dev)
        state["ssim"](gt[0], gt[0])
    except Exception:                                               # no convolution backend on this box
        state["ssim_ok"] = False
    pyrng = random.Random(seed)

    ssp_leaf = torch.zeros((P, 3), device=dev, requires_grad=True)

    def body(cc, ct, bg_map, gt_v, tt, degree, net_on):
        """The device work of one iteration on the given camera pair, background, ground truth and time tensor."""
        d_xyz, d_rot, d_sh, d_sh_p = 0.0, 0.0, 0.0, 0.0
        if net_on:                                                   # gaussian_model.py:170-174
            d_xyz, d_rot, d_sh, d_sh_p = net(par["xyz"].detach(), tt, zeros_as_scalars=True)      # (0.0 for the two all-zero offsets, as line above)
        # screenspace_points (gaussian_renderer/__init__.py:52-56): zeros whose only role is to receive a gradient -- the same
        # leaf every iteration (its values are never written), the last gradient dropped
        ssp = ssp_leaf
        ssp.grad = None
        if fused_params:
            # pc.get_* (gaussian_model.py:123-153: exp, normalize, sigmoid, the concatenations of the feature tensors) inside the
            # assembly's kernels, forward and backward: gftorf_amd.assemble_parameters
            m3, m2, op, sc, ro, shs, shp = assemble_parameters(
                par["xyz"], ssp, par["opacity"], par["scaling"], par["rotation"], par["f_dc_color"], par["f_rest_color"], par["phase_f_dc"],
                par["phase_f_rest"], par["amp_f_dc"], par["amp_f_rest"], mask, d_xyz, d_rot, d_sh, d_sh_p, render_regions=("dynamic",))
        else:
            # activations (gaussian_model.py:123-153), eagerly as the reference has them
            scaling = torch.exp(par["scaling"])
            rotation = torch.nn.functional.normalize(par["rotation"])
            opacity = torch.sigmoid(par["opacity"])
            feat_c = torch.cat((par["f_dc_color"], par["f_rest_color"]), dim=1)
            feat_p = torch.cat((torch.cat((par["phase_f_dc"], par["phase_f_rest"]), dim=1),
                                torch.cat((par["amp_f_dc"], par["amp_f_rest"]), dim=1)), dim=2)
            m3, m2, op, sc, ro, shs, shp = assemble_inputs(par["xyz"], ssp, opacity, scaling, rotation, par["rotation"], feat_c, feat_p,
                                                          mask, d_xyz, d_rot, d_sh, d_sh_p, render_regions=("dynamic",))
        if pair:      # opt-in: both calls as one node (gftorf_amd/pair.py); same outputs
            out_c, out_t = GaussianRasterizerPair(settings(cc, bg_map, degree, False), settings(ct, bg_map, degree, True))(
                means3D=m3, means2D=m2, opacities=op, shs=shs, shs_p=shp, scales=sc, rotations=ro,
                phase_offset=(0.0, phase_offset), dc_offset=(0.0, 0.0))
        else:
            out_c = GaussianRasterizer(settings(cc, bg_map, degree, False))(
                means3D=m3, means2D=m2, opacities=op, shs=shs, shs_p=shp, scales=sc, rotations=ro)
            out_t = GaussianRasterizer(settings(ct, bg_map, degree, True))(
                means3D=m3, means2D=m2, opacities=op, shs=shs, shs_p=shp, scales=sc, rotations=ro, phase_offset=phase_offset,
                dc_offset=0.0)
        tof = out_t[1][:cfg["num_phasor_channels"]]
        if fused_loss:
            # lambda_tof * ((1 - lambda_dssim) * l2 + lambda_dssim * (1 - ssim)) as one tensor (train.py:196-231)
            loss = gft_loss.weighted_loss(tof, gt_v, cfg["lambda_tof"] * (1.0 - cfg["lambda_dssim"]), cfg["lambda_tof"] * cfg["lambda_dssim"])
        else:
            l2 = ((tof - gt_v) ** 2).mean()
            loss = cfg["lambda_tof"] * ((1.0 - cfg["lambda_dssim"]) * l2 +
                                        (cfg["lambda_dssim"] * (1.0 - state["ssim"](tof, gt_v)) if state["ssim_ok"] else 0.0))
        # (render() always makes the colour-camera call, gaussian_renderer/__init__.py:107; with lambda_color = 0 its
        # outputs reach no loss, train.py:206, so autograd never runs its backward -- as in the reference)
        del out_c
        loss.backward()
        with torch.no_grad():
            radii = out_t[10]
            densify.add_densification_stats(stats[0], stats[1], stats[2], ssp.grad, radii > 0, out_t[8], radii)
            opt.step()
            opt.zero_grad(set_to_none=True)
            if net_on:
                opt_net.step()
                opt_net.zero_grad(set_to_none=True)
        return loss.detach()

    # ---- graph mode: what an iteration reads that changes from one to the next, as static tensors
    if graph:
        K = 80       # floats per view: colour camera {view 16, proj 16, campos 3 + 1}, ToF camera the same, time, padding
        table = torch.zeros((V, K), device=dev)
        for v, (cc, ct) in enumerate(cams):
            for o, cam in ((0, cc), (36, ct)):
                table[v, o:o + 16] = cam["_view"].reshape(-1)
                table[v, o + 16:o + 32] = cam["_proj"].reshape(-1)
                table[v, o + 32:o + 35] = cam["_campos"].reshape(-1)
            table[v, 72] = v / (V - 1)
        s_row = torch.zeros((K,), device=dev)
        s_gt = torch.zeros_like(gt[0])
        s_bg = torch.zeros((7, H, W), device=dev)
        gt_all = torch.stack(gt).contiguous()
        static_cams = []
        for o, cam in ((0, cams[0][0]), (36, cams[0][1])):               # (the views share their intrinsics)
            sc_ = dict(cam)
            sc_["_view"], sc_["_proj"], sc_["_campos"] = s_row[o:o + 16].view(4, 4), s_row[o + 16:o + 32].view(4, 4), s_row[o + 32:o + 35]
            static_cams.append(sc_)
        s_tt = s_row[72:73].view(1, 1).expand(P, -1)
        graphs = {}          # (degree, network on) -> [eager iterations so far, CUDAGraph or None, static loss]
        import os
        EAGER_FIRST = int(os.environ.get("GFT_LOOP_EAGER_FIRST", "3"))      # iterations of a configuration that run eagerly before it is captured

    def iteration(it):
        for gr in opt.param_groups:                                  # gaussian_model.py:294-310
            if gr["name"] == "xyz":
                gr["lr"] = xyz_lr(it)
        opt_net.param_groups[0]["lr"] = net_lr(it - cfg["warm_up"])
        if it % 1000 == 0 and state["degree"] < cfg["sh_degree"]:
            state["degree"] += 1
        if not state["stack"]:
            state["stack"] = list(range(V))
        v = state["stack"].pop(pyrng.randint(0, len(state["stack"]) - 1))
        net_on = it > cfg["warm_up"]
        # background seeded by the iteration, generator state kept (train.py:120-129)
        rs = torch.random.get_rng_state()
        torch.manual_seed(it)
        if not graph:
            bg_map = torch.empty((7, H, W), dtype=torch.float32, device=dev).uniform_(-1.0, 1.0)      # = torch.rand(...) * 2 - 1, bit for bit, in one launch
            torch.random.set_rng_state(rs)
            tt = torch.full((1, 1), v / (V - 1), device=dev).expand(P, -1) if net_on else None
            cc, ct = cams[v]
            return body(cc, ct, bg_map, gt[v], tt, state["degree"], net_on)
        s_bg.uniform_(-1.0, 1.0)
        torch.random.set_rng_state(rs)
        s_row.copy_(table[v])
        s_gt.copy_(gt_all[v])
        key = (state["degree"], net_on)
        g = graphs.setdefault(key, [0, None, None])
        if g[1] is None:
            if g[0] < EAGER_FIRST:
                g[0] += 1
                return body(static_cams[0], static_cams[1], s_bg, s_gt, s_tt, state["degree"], net_on)
            # torch's recipe for a whole-iteration capture: no gradient tensors alive, capture on a side stream
            torch.cuda.synchronize()
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                g[2] = body(static_cams[0], static_cams[1], s_bg, s_gt, s_tt, state["degree"], net_on)
            g[1] = cg
            # (the capture ran nothing: this iteration is the first replay)
        opt.refresh_lr()
        opt_net.refresh_lr()
        g[1].replay()
        return g[2]

    info = dict(P=P, W=W, H=H, views=V, ssim=state["ssim_ok"], activations="gftorf_amd.assemble_parameters (inside the assembly)" if fused_params else "torch: exp / sigmoid / normalize / cat, eagerly", frame=dict(cams=cams, g0=g0), par=par, net=net, opt=opt, opt_net=opt_net,
                graphs=(lambda: {k: v[1] is not None for k, v in graphs.items()}) if graph else (lambda: {}))
    return iteration, info


def run(dev, iters, sync, timed_region, cfg=C3, pair=False, graph=False, fused_loss=True):
    """Runs `iters` iterations inside timed_region(step_fn, n) -> seconds.  Returns (seconds, report dict)."""
    import torch
    iteration, info = build_loop(dev, cfg, pair=pair, graph=graph, fused_loss=fused_loss)
    counter = dict(it=0)
    losses = []

    def step():
        counter["it"] += 1
        l = iteration(counter["it"])
        if counter["it"] % 500 == 0 or counter["it"] == 1:
            losses.append((counter["it"], l.clone()))        # (graph mode returns one static tensor every iteration)
    secs = timed_region(step, iters)
    rep = dict(iterations=counter["it"], ssim_in_loss=info["ssim"] or fused_loss,
               loss_terms="gftorf_amd.loss.ssim_l2 (one launch forward, one backward)" if fused_loss else "torch: conv2d SSIM + elementwise L2",
               activations=info["activations"],
               loss_trace=[(i, float(l.item())) for i, l in losses])
    if graph:
        rep["graphs_captured"] = {"degree %d, network %s" % (k[0], "on" if k[1] else "off"): v for k, v in info["graphs"]().items()}
    return secs, rep, info
