#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ToF Gaussian rasterizer.

Metric (BASELINE.json): train iters/sec of the rasterizer forward+backward, plus
Mpix/s, at 1 M Gaussians @ 640x480, ToF view (SH colour + SH phasor, degree 3).
One "step" = one forward + one backward of the hot path through the public
operator API (GaussianRasterizer -> autograd -> C ABI -> gfx950 kernels) on one
synthetic frame whose inputs are already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload metric|C2|C5|tiny]

N > 1 is launched by the driver as `python -m torch.distributed.run ... bench.py
--gpus N ...`: one rank per GPU, each rank renders its own frame (frames are
independent units: weak scaling, no data-path collective).  Rank 0 prints ONE JSON
line.  Also in the line: `roofline` for the dominant kernel (HIP events on the
launch stream, algorithmic bytes of SURVEY.md 8(d)) and `cpu_baseline` (the CPU
oracle timed on this box's host cores; rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6290 GB/s measured copy ceiling

WORKLOADS = {
    "metric": dict(P=1_000_000, W=640, H=480, D=3, sh_coeffs=16, tof=True,
                   label="1M Gaussians, 640x480, SH deg 3, RGB+ToF phasor, forward+backward"),
    "C2": dict(P=500_000, W=640, H=480, D=3, sh_coeffs=16, tof=True,
               label="500k Gaussians, 640x480, SH deg 3, RGB+ToF phasor, forward+backward"),
    "C5": dict(P=5_000_000, W=1920, H=1080, D=3, sh_coeffs=16, tof=True,
               label="5M Gaussians, 1920x1080, SH deg 3, RGB+ToF phasor, forward+backward"),
    # the metric frame in the regime the reference trains in: opacities 0.05-0.1 (arguments/__init__.py:99 starts every
    # Gaussian at 0.1), so no pixel saturates, every tile list is walked whole and most visible Gaussians are blended
    "fog": dict(P=1_000_000, W=640, H=480, D=3, sh_coeffs=16, tof=True, opacity_range=(0.05, 0.1),
                label="fog: 1M Gaussians with opacity 0.05-0.1 (nothing saturates), 640x480, SH deg 3, RGB+ToF phasor, forward+backward"),
    # ... and the same regime at the largest shape: 8160 tiles with lists of ~9000 entries, every one walked whole
    "C5fog": dict(P=5_000_000, W=1920, H=1080, D=3, sh_coeffs=16, tof=True, opacity_range=(0.05, 0.1),
                  label="C5 fog: 5M Gaussians with opacity 0.05-0.1 (nothing saturates), 1920x1080, SH deg 3, RGB+ToF phasor, forward+backward"),
    "clustered": dict(P=1_000_000, W=640, H=480, D=3, sh_coeffs=16, tof=True, cluster=0.4,
                      label="1M Gaussians concentrated at the image centre, 640x480 (load-balance check)"),
    "tiny": dict(P=20_000, W=256, H=256, D=3, sh_coeffs=16, tof=True,
                 label="20k Gaussians, 256x256 (plumbing check only)"),
    # BASELINE.json config 1: the shape of the reference's CPU-runnable plumbing case, here forward-only on the GPU
    # with the CPU oracle timed beside it
    "C1": dict(P=10_000, W=256, H=256, D=0, sh_coeffs=1, tof=False, forward_only=True,
               label="10k Gaussians, 256x256, SH deg 0, RGB-only forward"),
    # BASELINE.json config 4: 8 frames x 1 M Gaussians, one frame per GPU.  A step is the per-rank iteration of the
    # reference's dynamic branch (train.py:164-178): deformation-network query at the frame's time, input assembly,
    # rasterizer forward + backward, network backward, ONE all-reduce of the network's gradient bucket (c4_step_fn)
    "C4": dict(P=1_000_000, W=640, H=480, D=3, sh_coeffs=16, tof=True, composed=True,
               label="C4: 8 frames x 1M Gaussians (30% dynamic), 640x480, one frame per GPU: deform query(t_g) + input assembly + "
                     "raster forward+backward + network backward + deform-gradient all-reduce per step"),
    # BASELINE.json config 3: handled by bench_loop.py (the whole optimisation loop, not one rasterizer call)
    "C3": dict(loop=True),
}


# ---------------------------------------------------------------------------
# distributed plumbing (device agnostic; exercised on CPU/gloo by tests/test_dist_gloo.py)
# ---------------------------------------------------------------------------
def dist_env():
    return dict(rank=int(os.environ.get("RANK", "0")), local_rank=int(os.environ.get("LOCAL_RANK", "0")),
                world=int(os.environ.get("WORLD_SIZE", "1")))


def frame_of_rank(rank, step=0, world=1):
    """Frames (views / time steps) are the independent units of this path: rank r owns
    frame step*world + r.  The bench re-renders the rank's frame every step."""
    return step * world + rank


def spin_up(step_fn, sync_fn, seconds=0.3):
    """Untimed steps until `seconds` of wall time have gone by on a busy device (synchronised every few steps, so the
    queue stays short): the power management raises the clocks over the first tens of milliseconds of load -- a leg that
    follows seconds of host-side scene building starts on an idle card (fog extra: 1.61 ms per step over 50 steps behind 15
    warm-up steps, 1.34 behind this; steps 2..25 of that leg went 1.67 -> 1.43 ms one by one)."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            step_fn()
        sync_fn()


def timed_steps(step_fn, steps, warmup, sync_fn, dist=None):
    """W untimed warm-up steps, then exactly K timed steps bracketed by barrier + device
    synchronisation on both sides.  Returns the elapsed seconds, MAX over ranks."""
    import torch
    for _ in range(warmup):
        step_fn()
    sync_fn()
    if dist is not None:
        dist.barrier()
    sync_fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync_fn()
    if dist is not None:
        dist.barrier()
    sync_fn()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


# ---------------------------------------------------------------------------
# algorithmic bytes; P_vis = radii > 0, R = tile instances, N = pixels
# ---------------------------------------------------------------------------
def algorithmic_bytes(P, P_vis, R, N, T, forward_only=False, units=None):
    """per visible Gaussian 1516 B (K1 484, K2 8, K3 20, K8 92, K9 836, accumulator clears 76), culled
    432 B, per instance 268 B (44 B of binning: keys 12, sort 24, ranges 8; 224 B of rendering: K6 76, K7 148), per
    pixel 224 B.  Without ``units`` this is the formula as written (the reference algorithm: every instance binned and
    rendered, every visible Gaussian given its appearance and a gradient row).

    With ``units`` the same per-unit constants are charged for the units this implementation's launches PROCESS; the
    lazy stages make the formula as written an over-count, not a bound (it printed > 8 TB/s at 5 M @ 1080p):
      P_app   visible Gaussians whose appearance is evaluated (SH colour + phasor: 320 B of SH read, 52 B written; the
              others cost K1 112 B of geometry) -- those that stand in the sorted part of some tile list
      R_bin   ids that are sorted into tile lists (the list heads, + the lists completed for flagged quadrants)
      R_walk  list entries the render stages walk (per tile up to its deepest contributor: early termination)
      P_blend Gaussians some pixel blended: K8 + K9 (928 B) run for those, every other Gaussian only gets its 376 B
              of zero gradient rows (+ 8 B of radii), like a culled one
      P_zero  Gaussians whose 376 B of zero gradient rows are written in this step (default: all but the blended ones;
              0 when the operator reuses gradient tensors it kept zero but for the rows of the previous backward)
      P_rezero rows of kept gradient tensors that are zeroed again (the previous backward's blended Gaussians: 376 B each,
              + 1 B of row marks per Gaussian), booked under `memset`"""
    P_cull = P - P_vis
    u = dict(P_app=P_vis, R_bin=R, R_walk=R, P_blend=P_vis, P_zero=None, P_rezero=0)
    if units:
        u.update({k: v for k, v in units.items() if v is not None})
    P_app, R_bin, R_walk, P_blend = u["P_app"], u["R_bin"], u["R_walk"], u["P_blend"]
    P_zero = u["P_zero"] if u["P_zero"] is not None else P - P_blend
    P_rezero = u["P_rezero"]
    per_kernel = {
        "preprocess_fwd": 484 * P_app + 112 * (P_vis - P_app) + 48 * P_cull,
        "tile_count": 8 * P_vis + 8 * R_bin,        # reference scan (K2) + tile ranges (K5)
        "tile_scatter": 20 * P_vis + 12 * R_bin,    # reference duplicateWithKeys (K3)
        "tile_sort": 24 * R_bin,                    # reference key sort (K4), one read + one write of a pair
        "render_fwd": 76 * R_walk + 128 * N,
        "render_bwd": 148 * R_walk + 96 * N,
        "preprocess_bwd": 928 * P_blend + 376 * P_zero + 8 * (P - P_blend),
        "memset": 76 * P_vis + (376 * P_rezero + P if P_rezero else 0),
    }
    if forward_only:       # 512 P_vis + 48 P_cull + 120 R + 128 N
        for k in ("render_bwd", "preprocess_bwd", "memset"):
            per_kernel[k] = 0
    return per_kernel, sum(per_kernel.values())


def walked_instances(dev, radii=None):
    """List entries the render stages have to touch: per tile, the deepest contributor over its four 8x8
    quadrants (the backward starts there; the forward stops a little later, when its last pixel saturates),
    read from the scratch of the most recent forward."""
    import torch
    from gftorf_amd import _lib, api
    b = api.last_call_buffers
    if not b or b.get("P", 0) == 0:
        return None
    L = _lib.get_layout(b["P"], b["W"], b["H"], b["cap"])
    T = ((b["W"] + 15) // 16) * ((b["H"] + 15) // 16)
    tm = b["img"][L.img_tile_max:L.img_tile_max + T * 16].view(torch.int32).reshape(T, 4)
    ctrl = b["img"][L.img_ctrl:L.img_ctrl + 64].view(torch.int32).cpu().tolist()
    need = b["geom"][L.geom_need:L.geom_need + b["P"]]
    heads = b["img"][L.img_front_len:L.img_front_len + 4 * T].view(torch.int32)
    # lists that hinted tiles sorted whole: in the pool (behind the T head slots of 2048 ids) and without a tail
    rng0 = b["img"][L.img_ranges:L.img_ranges + 8 * T].view(torch.int32).reshape(T, 2)[:, 0].long() & 0xffffffff
    cut = b["img"][L.img_tile_cut:L.img_tile_cut + 4 * T].view(torch.int32)
    whole_ids = int(heads[(rng0 >= T * 2048) & (cut == -1)].sum().item())
    pull = bool(ctrl[5])          # (Gaussian, supertile) entries: only the tile-pull count pass writes them
    return {"per_tile_deepest": int(tm.max(dim=1).values.sum().item()), "per_quadrant_sum": int(tm.sum().item()),
            "tile_pull": pull,
            # Gaussians that stand in the sorted part of some tile list and were given an appearance; None = all visible ones
            "gaussians_with_appearance": int((need != 0).sum().item()) if pull else None,
            # bookkeeping of that forward (ctrl words, gft_internal.h): entries dealt to supertiles, ids sorted into list
            # heads, quadrants that walked past their head, ids in the lists that were completed for them
            # (the lists that hinted tiles sorted whole lie in the pool like the completed ones and are counted in ctrl[6] as well
            # as in the tiles' sorted lengths: taken out of the completed ones here)
            "supertile_entries": ctrl[5] if pull else None, "head_ids": int(heads.sum().item()) if pull else None,
            "flagged_quadrants": ctrl[4], "completed_list_ids": (ctrl[6] - whole_ids) if pull else None,
            "whole_list_ids": whole_ids if pull else None}


def build_scene(workload, rank, world):
    from gftorf_amd import synth
    cfg = dict(WORKLOADS[workload])
    frame = frame_of_rank(rank, 0, world)
    # one frame per rank: same Gaussian population, its own view (small arc); the per-frame deform offsets of the
    # reference's dynamic scenes are part of the composed workload (C4: c4_step_fn), not of the raster-only ones
    w2c = synth.look_at_w2c(yaw=0.02 * frame, pitch=-0.01 * frame, t=(0.01 * frame, 0.0, 0.0))
    scene = synth.make_scene(cfg, seed=1234, w2c=w2c)
    scene["label"] = cfg["label"]
    return scene


def gpu_step_fn(scene, dev):
    import numpy as np
    import torch
    from gftorf_amd import GaussianRasterizationSettings, GaussianRasterizer
    cam, cfg, g = scene["cam"], scene["cfg"], scene["gaussians"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    settings = GaussianRasterizationSettings(
        image_height=cfg["H"], image_width=cfg["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
        bg=t(scene["bg"]), scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]),
        projmatrix=t(cam["projmatrix"]), sh_degree=cfg["D"], campos=t(cam["campos"]), prefiltered=False,
        debug=False, near_n=cam["znear"], far_n=cam["zfar"], depth_range=scene["depth_range"],
        use_view_dependent_phase=scene["use_view_dependent_phase"])
    rast = GaussianRasterizer(settings)
    leaf = {k: t(v).requires_grad_(True) for k, v in g.items() if v is not None}
    means2D = torch.zeros((cfg["P"], 3), device=dev, requires_grad=True)
    gr = {k: t(v) for k, v in scene["grads"].items()}
    state = {}

    def step():
        for v in leaf.values():
            v.grad = None
        means2D.grad = None
        if cfg.get("forward_only"):
            with torch.no_grad():
                outs = rast(means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"], shs=leaf["shs"],
                            shs_p=leaf.get("shs_p"), scales=leaf["scales"], rotations=leaf["rotations"],
                            phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"])
            state["radii"] = outs[10]
            state["pixels"] = outs[8]
            return
        outs = rast(means3D=leaf["means3D"], means2D=means2D, opacities=leaf["opacities"], shs=leaf["shs"],
                    shs_p=leaf.get("shs_p"), scales=leaf["scales"], rotations=leaf["rotations"],
                    phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"])
        color, phasor, depth, _, acc, _, dd = outs[:7]
        # fixed random upstream gradients: every differentiable output is exercised
        torch.autograd.backward([color, phasor, depth, acc, dd],
                                [gr["color"], gr["phasor"], gr["depth"], gr["acc"], gr["depth_distortion"]])
        state["radii"] = outs[10]
        state["pixels"] = outs[8]

    return step, state, leaf


def c4_step_fn(scene, dev, dist, rank, world):
    """BASELINE.json config 4 as one composed step per rank (gftorf_amd.frames.FrameStep): rank r renders frame
    step*world + r of 8 -- its own camera, its own time t_g for the deformation network --, and the step ends with the
    path's one collective, the all-reduce of the network's gradient bucket.  Same Gaussians and network replica on
    every rank; no optimizer in the step (the metric is the rasterizer's forward + backward)."""
    import numpy as np
    import torch
    from gftorf_amd import GaussianRasterizationSettings, GaussianRasterizer, reference_network
    from gftorf_amd.frames import FrameStep
    cam, cfg, g = scene["cam"], scene["cfg"], scene["gaussians"]
    P = cfg["P"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    settings = GaussianRasterizationSettings(
        image_height=cfg["H"], image_width=cfg["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(scene["bg"]),
        scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=cfg["D"],
        campos=t(cam["campos"]), prefiltered=False, debug=False, near_n=cam["znear"], far_n=cam["zfar"],
        depth_range=scene["depth_range"], use_view_dependent_phase=scene["use_view_dependent_phase"])
    rast = GaussianRasterizer(settings)
    rng = np.random.default_rng(21)                         # same dynamic set and same replica on every rank
    mask = torch.tensor(rng.random(P) < 0.3, device=dev)
    torch.manual_seed(7)
    net = reference_network()
    for name, p in net.named_parameters():
        torch.nn.init.normal_(p, 0.0, 0.06 if (name.startswith("linear") and name.endswith("weight")) else 1e-3)
    net = net.to(dev)
    leaf = dict(xyz=t(g["means3D"]), opacity=t(g["opacities"]).reshape(P, 1), scaling=t(g["scales"]),
                rotation_raw=t(g["rotations"]), fc=t(g["shs"]), fp=t(g["shs_p"]))
    for v in leaf.values():
        v.requires_grad_(True)
    gr = scene["grads"]
    upstream = [t(gr[k]) for k in ("color", "phasor", "depth", "acc", "depth_distortion")]

    def render(frame_id, **kw):
        return rast(phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"], **kw)
    fs = FrameStep(net, leaf, mask, render, upstream, dist=dist, num_frames=8)
    state = {"it": 0, "frame_step": fs}

    def step():
        fs.zero_grad()
        outs = fs(frame_of_rank(rank, state["it"], world))
        state["it"] += 1
        state["radii"], state["pixels"] = outs[10], outs[8]
    return step, state, leaf


def cpu_baseline(scene, budget_s=20.0, max_iters=5, forward_only=False, one_core=True, one_core_budget_s=35.0):
    """The CPU oracle (a port of the reference algorithm; the reference itself has no CPU
    rasterizer) timed on this host: full forward+backward of the same frame."""
    from oracle import oracle
    import helpers as Hh
    oracle.build()
    oracle.lib()
    # output / scratch arrays kept between iterations and zero-filled on all cores (as a CPU rasterizer would keep its
    # buffers; fresh numpy arrays cost single-threaded page-fault and unmap work per iteration)
    oracle.reuse_buffers(True)
    try:
        return _cpu_baseline(oracle, Hh, scene, budget_s, max_iters, forward_only, one_core, one_core_budget_s)
    finally:
        oracle.reuse_buffers(False)


def _cpu_baseline(oracle, Hh, scene, budget_s, max_iters, forward_only, one_core, one_core_budget_s):
    t_all = []
    default_threads = oracle.num_threads()
    oracle.set_num_threads(oracle.cpu_budget())        # the CPUs this process may really use (cgroup quota), not the host's count
    t_start = time.perf_counter()
    Hh.run_oracle(oracle, scene, backward=not forward_only)  # warm-up (page faults, thread pool)
    warm = time.perf_counter() - t_start
    while len(t_all) < max_iters and (time.perf_counter() - t_start) < budget_s:
        t0 = time.perf_counter()
        Hh.run_oracle(oracle, scene, backward=not forward_only)
        t_all.append(time.perf_counter() - t0)
    if not t_all:
        t_all = [warm]
    t_all.sort()
    med = t_all[len(t_all) // 2]
    cores = oracle.num_threads()
    out = dict(value=1.0 / med, unit="it/s", cores=cores, kind="port",
               sample="%d full %s iterations of the same frame (median %.3f s) on %d OpenMP threads = the CPUs this "
                      "process may use (%d logical CPUs on the host)" % (
                   len(t_all), "forward" if forward_only else "forward+backward", med, cores, os.cpu_count() or 0))
    if one_core and cores > 1:
        # "with all cores ... and with 1 core".  One thread takes ~15 s per 1 M frame: a single full iteration
        # (a second one while the budget lasts), no warm-up run -- the pages are warm from the runs above
        oracle.set_num_threads(1)
        try:
            t1, t_start = [], time.perf_counter()
            while len(t1) < 2 and (not t1 or time.perf_counter() - t_start + t1[0] < one_core_budget_s):
                t0 = time.perf_counter()
                Hh.run_oracle(oracle, scene, backward=not forward_only)
                t1.append(time.perf_counter() - t0)
        finally:
            oracle.set_num_threads(cores)
        out["one_core"] = dict(value=1.0 / min(t1), unit="it/s", cores=1, kind="port",
                               sample="%d full iteration(s) of the same frame on one thread (best %.2f s)" % (len(t1), min(t1)))
    oracle.set_num_threads(default_threads)
    return out


def load_counters(stage, workload):
    """Counter summary of one stage from the committed rocprofv3 --pmc passes (profiles/collect_pmc.sh +
    make_counters.py -> profiles/counters.json): HBM bytes per launch (factor*FETCH_SIZE + WRITE_SIZE, the factor calibrated per
    access kind: profiles/fetch_factors.py), VALU
    wave-instructions per launch, resident waves per SIMD, VALU issue share.  None when the workload was not profiled."""
    path = os.path.join(ROOT, "profiles", "counters.json")
    try:
        with open(path) as f:
            d = json.load(f)
        w = d.get(workload)
        if not w or stage not in w["stages"]:
            return None
        s = dict(w["stages"][stage])
        s["source"] = w["source"]
        s["kernel_source_sha"] = w.get("kernel_source_sha")
        return s
    except Exception:
        return None


def kernel_source_sha():
    """sha1 over gftorf_amd/csrc/*.{hip,h}: the committed counter passes (profiles/counters.json) carry the hash of the
    sources they were taken from, so a line can say whether its `traffic` belongs to the kernels that ran."""
    import hashlib
    h = hashlib.sha1()
    root = os.path.join(ROOT, "gftorf_amd", "csrc")
    for f in sorted(os.listdir(root)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            with open(os.path.join(root, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


FP32_VALU_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: peak FP32 (vector)
# a (pixel, Gaussian) pair costs ~30 flop in the forward blend and ~150 flop in the backward
FLOPS_PER_PAIR = {"render_fwd": 30.0, "render_bwd": 150.0}


def assemble_extra(dev, P=1_000_000, frac=0.3, M=16, steps=20, warmup=5):
    """SURVEY 8(f) row 1 beside the headline metric: the fused input assembly (forward + backward)
    against the reference's eager-PyTorch glue (gaussian_renderer/__init__.py:81-105, restated in
    oracle/assemble_ref.py) on the same device.  Algorithmic bytes per Gaussian at M = 16, fp32:
    forward reads 12+12+4+12+16+192+128 = 376 (+ 348 of offsets for a dynamic row) and writes 388;
    backward reads 388 and writes 392 (+ 348 for a dynamic row) -- since round 6 without the 320 bytes of SH gradient, which the
    backward hands on as they are."""
    import numpy as np
    import torch
    from gftorf_amd import assemble_inputs
    from oracle import assemble_ref
    rng = np.random.default_rng(99)
    f = lambda *s: torch.tensor(rng.normal(0, 1, s).astype(np.float32), device=dev)
    mask = torch.tensor(rng.random(P) < frac, device=dev)
    nd = int(mask.sum().item())
    raw = f(P, 4)
    src = [f(P, 3), f(P, 3), torch.rand((P, 1), device=dev), f(P, 3).exp(), torch.nn.functional.normalize(raw), raw,
           f(P, M, 3), f(P, M, 2)]
    offs = [f(nd, 3), f(nd, 4), f(nd, M, 3), f(nd, M, 2)]
    for t in src + offs:
        t.requires_grad_(True)
    gout = None

    class _FromRasterizer(torch.autograd.Function):
        # The gradients reach the assembly as the rasterizer's backward hands them over: tensors nobody else holds, which
        # autograd may keep as a leaf's .grad.  (Gradients the caller holds -- a plain backward(outs, gout) -- are cloned
        # by autograd on their way into .grad: 320 bytes per Gaussian of torch's, not of the op's.)
        @staticmethod
        def forward(ctx, x, g):
            ctx.g = g
            return x.view_as(x)

        @staticmethod
        def backward(ctx, _):
            return ctx.g.view_as(ctx.g), None

    def step(fn):
        nonlocal gout
        outs = fn(*src, mask, *offs)
        if gout is None:
            gout = [torch.randn_like(o) for o in outs]
        torch.autograd.backward([_FromRasterizer.apply(o, g) for o, g in zip(outs, gout)], gout)
        for t in src + offs:
            t.grad = None

    def timed(fn, n, w):
        for _ in range(w):
            step(fn)
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n):
            step(fn)
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / n

    fused_ms = timed(assemble_inputs, steps, warmup)
    eager_ms = timed(assemble_ref.assemble_eager, max(3, steps // 4), 2)
    # (round 6: the backward hands the SH rows' gradient on instead of copying it -- 320 bytes per Gaussian neither read nor
    # written again; the dynamic rows are read once more for d_sh / d_sh_p)
    bytes_alg = P * (376 + 388 + (388 - 320) + (392 - 320)) + nd * (348 + 348 + 320)
    return {"what": "fused input assembly fwd+bwd (SURVEY 8(f)#1), %d Gaussians, %d dynamic, SH 16" % (P, nd),
            "fused_ms": fused_ms, "eager_torch_ms": eager_ms, "speedup_vs_eager": eager_ms / fused_ms,
            "algorithmic_bytes": bytes_alg, "achieved_GBs": bytes_alg / (fused_ms * 1e-3) / 1e9,
            "frac_of_hbm_peak": bytes_alg / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense fp32 in / fp32 accumulate
BF16_MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA; a product of the deformation network costs six of them


def deform_extra(dev, n=300_000, steps=10, warmup=3):
    """SURVEY 8(f) row 2 beside the headline metric: one query of the deformation network for the dynamic
    Gaussians of the metric frame (30 % of 1 M), forward + backward, against the eager-torch statements
    of the reference's module (utils/time_utils.py:103-127, restated in oracle/deform_ref.py) on the same
    device.  The network is the one the reference constructs (scene/deform_model.py:9-16: t_multires 10, 84
    encoded inputs).  Algorithmic multiply-adds per point: forward 84*256 + 6*256*256 + 340*256 + 51*256 =
    514 816; backward 7*256*256 + 51*256 (activation gradients) + 514 816 (weight gradients).  The roofline is the
    matrix pipe at six bf16 MFMAs per fp32 product (2 500 / 6 = 417 TFLOP/s of fp32 products)."""
    import numpy as np
    import torch
    from gftorf_amd import reference_network, _lib
    from gftorf_amd import synth
    from oracle import deform_ref           # the eager / CPU baseline legs below; the HIP leg's inputs come from synth
    params = synth.random_deform_params(3)
    net = reference_network()
    net.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    net = net.to(dev)
    pt = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in params.items()}
    rng = np.random.default_rng(4)
    x = torch.tensor(rng.random((n, 3)).astype(np.float32), device=dev)
    t = torch.full((1, 1), 0.4, device=dev).expand(n, -1)          # scene/gaussian_model.py:171
    g_dxyz, g_dsh = torch.randn((n, 3), device=dev), torch.randn((n, 16, 3), device=dev)

    def ours():
        d_xyz, _, d_sh, _ = net(x, t)
        torch.autograd.backward([d_xyz, d_sh], [g_dxyz, g_dsh])
        net.zero_grad(set_to_none=True)

    def ours_fwd():
        with torch.no_grad():
            net(x, t)

    def eager():
        d_xyz, _, d_sh, _ = deform_ref.deform_eager(pt, x, t)
        torch.autograd.backward([d_xyz, d_sh], [g_dxyz, g_dsh])
        for p in pt.values():
            p.grad = None

    def timed(fn, k, w):
        for _ in range(w):
            fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / k

    ms = timed(ours, steps, warmup)
    fwd_ms = timed(ours_fwd, steps, warmup)
    eager_ms = timed(eager, max(3, steps // 2), 2)
    # CPU beside it: the numpy oracle (fp32 BLAS on the host cores), forward + backward, on a bounded sample
    import os
    import time
    ns = 20_000
    xs, ts = x[:ns].cpu().numpy(), np.full((ns, 1), 0.4, np.float32)
    gx, gs = g_dxyz[:ns].cpu().numpy(), g_dsh[:ns].cpu().numpy()
    deform_ref.backward(params, xs[:2000], ts[:2000], gx[:2000], gs[:2000])
    t0 = time.perf_counter()
    deform_ref.backward(params, xs, ts, gx, gs)
    cpu_s = time.perf_counter() - t0
    n_in = net.xyz_input_ch + net.t_input_ch                     # 84
    macs_fwd = n_in * 256 + 6 * 256 * 256 + (n_in + 256) * 256 + 51 * 256
    macs = macs_fwd + (7 * 256 * 256 + 51 * 256) + macs_fwd
    tf = 2.0 * macs * n / (ms * 1e-3) / 1e12
    tf_fwd = 2.0 * macs_fwd * n / (fwd_ms * 1e-3) / 1e12
    # matrix-pipe multiplies per fp32 product: the forward walk takes 3 (two fp16 planes per operand; GFT_DEFORM_FP16X2=0: 6,
    # three bf16 planes); since round 6 so do the backward walk and the hidden layers' weight gradients
    # (GFT_DEFORM_BWD_FP16=0: 6); the first layer's, the skip rows' and the heads' weight gradients take 6
    fwd_terms = 3.0 if os.environ.get("GFT_DEFORM_FP16X2", "1") != "0" else 6.0
    bwd_terms = 3.0 if os.environ.get("GFT_DEFORM_BWD_FP16", "1") != "0" else 6.0
    peak_fwd = BF16_MFMA_PEAK_TFLOPS / fwd_terms
    macs_light = 2 * n_in * 256 + 51 * 256                      # dW of layer 0, of layer 5's encoding rows, of the heads
    macs_bwd_h = (macs - macs_fwd) - macs_light                 # the backward walk + the hidden layers' dW
    peak_all = macs / (macs_fwd / peak_fwd + macs_bwd_h / (BF16_MFMA_PEAK_TFLOPS / bwd_terms) + macs_light / (BF16_MFMA_PEAK_TFLOPS / 6.0))
    return {"what": "deformation network fwd+bwd (SURVEY 8(f)#2), %d points, fp32 results from the 16-bit matrix pipe: three "
                    "fp16 MFMAs per product in the forward walk, the backward walk and the hidden layers' weight gradients, six "
                    "bf16 MFMAs in the other weight gradients (GFT_DEFORM_FP16X2=0 / GFT_DEFORM_BWD_FP16=0: six bf16 ones in the "
                    "forward / backward; GFT_DEFORM_BF16X3=0: fp32-operand MFMA)" % n,
            "fwd_bwd_ms": ms, "inference_fwd_ms": fwd_ms, "eager_torch_ms": eager_ms, "speedup_vs_eager": eager_ms / ms,
            "algorithmic_flops": 2.0 * macs * n, "achieved_TFLOPs": tf, "inference_TFLOPs": tf_fwd,
            "architecture": dict(D=net.D, W=net.W, xyz_multires=net.xyz_multires, t_multires=net.t_multires,
                                 encoded_inputs=n_in, parameters=sum(p.numel() for p in net.parameters())),
            "roofline": {"bound": "mfma", "achieved": tf, "peak": peak_all, "unit": "TFLOP/s", "frac": tf / peak_all,
                         "note": "fp32 products on the 16-bit matrix pipe (2500 TFLOP/s dense): %d multiplies each in the forward "
                                 "walk, %d in the backward walk and the hidden layers' weight gradients, 6 in the other "
                                 "weight gradients; peak = the flop-weighted mix" % (int(fwd_terms), int(bwd_terms))},
            "inference_frac": tf_fwd / peak_fwd, "inference_peak_TFLOPs": peak_fwd,
            "ratio_to_fp32_operand_mfma_peak": tf / FP32_MFMA_PEAK_TFLOPS,
            "points_per_s": n / (ms * 1e-3),
            "cpu_baseline": {"value": ns / cpu_s, "unit": "points/s", "cores": os.cpu_count(), "kind": "port",
                             "sample": "%d points, forward + backward of oracle/deform_ref.py (numpy fp32 BLAS)" % ns}}


def deform_exchange(dev, dist, rank, world, n=300_000, steps=10, warmup=3):
    """The path's one exchange step, measured on all ranks beside the sharded raster metric:
    every rank queries the deformation network for its own frame's time (fwd + bwd, 30 % of 1 M Gaussians),
    then ONE all-reduce of the 2.07 MB gradient bucket (RCCL over xGMI) makes the replicas' gradients equal.
    Barrier-bracketed, MAX over ranks, like the headline timing."""
    import numpy as np
    import torch
    from gftorf_amd import reference_network
    from gftorf_amd.deform import allreduce_gradients, flat_grad_bucket
    torch.manual_seed(7)                                  # identical replicas
    net = reference_network()
    for name, p in net.named_parameters():
        if name.endswith("weight"):
            torch.nn.init.normal_(p, 0.0, 0.06)
    net = net.to(dev)
    rng = np.random.default_rng(11)                       # same Gaussians on every rank, own frame time
    x = torch.tensor(rng.random((n, 3)).astype(np.float32), device=dev)
    g_dxyz, g_dsh = torch.randn((n, 3), device=dev), torch.randn((n, 16, 3), device=dev)
    nbytes = [0]

    def make_step(exchange):
        def step_fn():
            t = torch.full((1, 1), (rank + 0.5) / world, device=dev).expand(n, -1)
            d_xyz, _, d_sh, _ = net(x, t)
            torch.autograd.backward([d_xyz, d_sh], [g_dxyz, g_dsh])
            if exchange:
                nbytes[0] = allreduce_gradients(net, dist, average=True)
            else:
                flat_grad_bucket(net)                     # same bucket assembly, no exchange
        return step_fn

    def run(exchange):
        fn = make_step(exchange)

        def one():
            fn()
            net.zero_grad(set_to_none=True)
        return timed_steps(one, steps, warmup, lambda: torch.cuda.synchronize(dev), dist) / steps * 1e3

    local_ms = run(False)
    total_ms = run(True)
    # replicas must hold bit-identical gradients after the exchange
    make_step(True)()
    flat, _ = flat_grad_bucket(net)
    probe = torch.stack([flat.double().sum(), flat.double().abs().max()])
    on_dev = dist.get_backend() == "nccl"
    lo, hi = (probe.clone(), probe.clone()) if on_dev else (probe.cpu(), probe.cpu())
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    net.zero_grad(set_to_none=True)
    # the collective alone
    buf = torch.zeros(nbytes[0] // 4, device=dev if on_dev else "cpu")

    def only():
        dist.all_reduce(buf)
    coll_ms = timed_steps(only, 5 * steps, warmup, lambda: torch.cuda.synchronize(dev), dist) / (5 * steps) * 1e3
    return {"what": "deform network fwd+bwd on %d points per rank + all-reduce of the gradient bucket" % n,
            "backend": dist.get_backend(), "bucket_bytes": nbytes[0], "ms_per_step_no_exchange": local_ms,
            "ms_per_step": total_ms, "allreduce_alone_ms": coll_ms,
            "allreduce_busbw_GBs": 2.0 * (world - 1) / world * nbytes[0] / (coll_ms * 1e-3) / 1e9,
            "replicas_identical": bool(torch.equal(lo, hi))}


def densify_extra(dev, P=1_000_000):
    """SURVEY 8(f) row 4 (bookkeeping part) beside the headline metric: the per-iteration statistics update
    (train.py:441-449) and one pruning step over the model's 11 parameter tensors with their Adam moments
    and the 3 statistics tensors (scene/gaussian_model.py:473-514), against the reference's eager statements
    (oracle/densify_ref.py) on the same device.  Algorithmic bytes: statistics 1 + 33 per visible Gaussian
    (flag; xy gradient, pixels, radius, and three read-modify-writes); pruning: every row read once, every
    kept row written once, + mask and rank."""
    import numpy as np
    import torch
    from gftorf_amd import densify
    from oracle import densify_ref
    g = torch.Generator().manual_seed(12)
    f = lambda *s: torch.randn(s, generator=g).to(dev)
    vis = (torch.rand(P, generator=g) < 0.9).to(dev)
    grad, pixels = f(P, 3) * 1e-3, (torch.rand((P, 1), generator=g) * 300).floor().to(dev)
    radii = (torch.rand(P, generator=g) * 40).to(torch.int32).to(dev) * vis
    acc, den, mr = torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev), torch.zeros(P, device=dev)

    def timed(fn, k, w=2):
        for _ in range(w):
            fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / k

    stats_ms = timed(lambda: densify.add_densification_stats(acc, den, mr, grad, vis, pixels, radii), 20)
    stats_eager_ms = timed(lambda: densify_ref.add_densification_stats_eager(acc, den, mr, grad, vis, pixels, radii), 5)
    nvis = int(vis.sum().item())
    stats_bytes = P + 44 * nvis
    rows = [(3,), (1, 3), (15, 3), (1, 1), (15, 1), (1, 1), (15, 1), (1,), (3,), (4,), (3,)]
    tensors = [f(P, *r) for r in rows for _ in range(3)] + [acc, den, mr]        # parameter + two moments each
    keep = (torch.rand(P, generator=g) < 0.9).to(dev)
    prune_ms = timed(lambda: densify.select_rows(keep, *tensors), 5)
    prune_eager_ms = timed(lambda: [t[keep] for t in tensors], 3)
    nkeep = int(keep.sum().item())
    row_bytes = sum(t.numel() // P * 4 for t in tensors)
    prune_bytes = P * (row_bytes + 1 + 4 + 4) + nkeep * row_bytes
    return {"what": "per-Gaussian bookkeeping (SURVEY 8(f)#4), %d Gaussians" % P,
            "stats_ms": stats_ms, "stats_eager_torch_ms": stats_eager_ms, "stats_speedup": stats_eager_ms / stats_ms,
            "stats_GBs": stats_bytes / (stats_ms * 1e-3) / 1e9, "stats_frac_of_hbm_peak": stats_bytes / (stats_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "prune_tensors": len(tensors), "prune_ms": prune_ms, "prune_eager_torch_ms": prune_eager_ms,
            "prune_speedup": prune_eager_ms / prune_ms, "prune_GBs": prune_bytes / (prune_ms * 1e-3) / 1e9,
            "prune_frac_of_hbm_peak": prune_bytes / (prune_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}


def train_iteration_extra(dev, scene, steps=10, warmup=3, only_fused=False):
    """One iteration of the reference's loop shape (train.py:164-177, 441-449, 470-474) at the metric size: one
    deformation-network query for the dynamic 30 %, one input assembly, the colour-camera and the ToF-camera rasterizer
    call (forward + backward of both), densification statistics, Adam on the Gaussians and on the network.  `hip`: every piece
    from this package; `eager`: the same rasterizer with the reference's eager statements around it
    (oracle/*_ref.py restatements on the device, torch.optim.Adam).  There is no reference rasterizer for
    ROCm, so both columns share ours."""
    import numpy as np
    import torch
    from gftorf_amd import (reference_network, FusedAdam, GaussianRasterizationSettings, GaussianRasterizer,
                            GaussianRasterizerPair, assemble_inputs, densify)
    from oracle import assemble_ref, deform_ref, densify_ref
    cam, cfg, g = scene["cam"], scene["cfg"], scene["gaussians"]
    P = cfg["P"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    settings = GaussianRasterizationSettings(
        image_height=cfg["H"], image_width=cfg["W"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(scene["bg"]),
        scale_modifier=1.0, viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=cfg["D"],
        campos=t(cam["campos"]), prefiltered=False, debug=False, near_n=cam["znear"], far_n=cam["zfar"],
        depth_range=scene["depth_range"], use_view_dependent_phase=scene["use_view_dependent_phase"])
    rast = GaussianRasterizer(settings)
    rng = np.random.default_rng(21)
    mask = torch.tensor(rng.random(P) < 0.3, device=dev)
    from gftorf_amd import synth
    params = synth.random_deform_params(9, head_std=1e-3)
    gr = {k: t(v) for k, v in scene["grads"].items()}

    def build(fused, pair=False):
        leaf = dict(xyz=t(g["means3D"]), opacity=t(g["opacities"]).reshape(P, 1), scaling=t(g["scales"]),
                    rotation_raw=t(g["rotations"]), fc=t(g["shs"]), fp=t(g["shs_p"]))
        for v in leaf.values():
            v.requires_grad_(True)
        groups = [{"params": [v], "lr": 1e-5, "name": k} for k, v in leaf.items()]
        if fused:
            net = reference_network()
            net.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
            net = net.to(dev)
            opt, net_params = FusedAdam(groups, lr=0.0, eps=1e-15), list(net.parameters())
        else:
            net = {k: torch.tensor(v, device=dev, requires_grad=True) for k, v in params.items()}
            opt, net_params = torch.optim.Adam(groups, lr=0.0, eps=1e-15), list(net.values())
        opt_net = torch.optim.Adam(net_params, lr=1e-6, eps=1e-15)
        x_n = leaf["xyz"].detach()[mask]
        x_n = (x_n - x_n.min(0).values) / (x_n.max(0).values - x_n.min(0).values)
        stats = [torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev), torch.zeros(P, device=dev)]

        pair_rast = GaussianRasterizerPair(settings, settings) if pair else None

        def iteration():
            tt = torch.full((1, 1), 0.4, device=dev).expand(x_n.size(0), -1)
            # (this package's network hands its two all-zero offsets over as the scalar 0.0, which assemble_inputs takes like
            # train.py:164's -- no [n, 4] / [n, 16, 2] zero tensors filled, added and given a gradient)
            d = net(x_n, tt, zeros_as_scalars=True) if fused else deform_ref.deform_eager(net, x_n, tt)
            # render() (gaussian_renderer/__init__.py:81-128): one input assembly, then the colour-camera and the
            # ToF-camera rasterizer calls on the same tensors
            ssp = torch.zeros((P, 3), device=dev, requires_grad=True)
            rot = torch.nn.functional.normalize(leaf["rotation_raw"])
            args = (leaf["xyz"], ssp, leaf["opacity"], leaf["scaling"], rot, leaf["rotation_raw"], leaf["fc"], leaf["fp"],
                    mask) + tuple(d)
            m3, m2, op, sc, ro, shs, shp = assemble_inputs(*args) if fused else assemble_ref.assemble_eager(*args)
            kw = dict(means3D=m3, means2D=m2, opacities=op, shs=shs, shs_p=shp, scales=sc, rotations=ro)
            if pair:
                out_c, out_t = pair_rast(phase_offset=(0.0, scene["phase_offset"]), dc_offset=(0.0, scene["dc_offset"]), **kw)
            else:
                out_c = rast(**kw)
                out_t = rast(phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"], **kw)
            loss = ((out_c[0] * gr["color"]).sum() + (out_c[2] * gr["depth"]).sum() +
                    (out_t[1] * gr["phasor"]).sum() + (out_t[2] * gr["depth"]).sum())
            loss.backward()                                     # one backward for the summed losses (train.py:364-366)
            radii, pixels = out_t[10], out_t[8]
            if fused:
                densify.add_densification_stats(stats[0], stats[1], stats[2], ssp.grad, radii > 0, pixels, radii)
            else:
                densify_ref.add_densification_stats_eager(stats[0], stats[1], stats[2], ssp.grad, radii > 0, pixels, radii)
            opt.step(); opt_net.step()
            opt.zero_grad(set_to_none=True); opt_net.zero_grad(set_to_none=True)
        return iteration

    def timed(fn, k, w):
        for _ in range(w):
            fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(k):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / k

    from gftorf_amd import deform as deform_mod
    hip_ms = timed(build(True), steps, warmup)
    net_rows = deform_mod.backward_stats()
    if only_fused:          # (profiles/train_workload.py: the iteration of this package alone, for rocprofv3)
        return {"hip_ms": hip_ms, "network_backward_fraction": net_rows["points_processed"] / max(net_rows["points"], 1)}
    torch.cuda.empty_cache()
    deform_mod.sparse_backward = False
    try:
        dense_ms = timed(build(True), steps, warmup)      # the network's backward over all queried points (round 2's)
    finally:
        deform_mod.sparse_backward = True
    torch.cuda.empty_cache()
    pair_ms = timed(build(True, pair=True), steps, warmup)
    torch.cuda.empty_cache()
    eager_ms = timed(build(False), max(3, steps // 2), 2)
    return {"what": "one training iteration of the reference's loop shape, %d Gaussians (30 %% dynamic), %dx%d: network query, "
                    "input assembly, colour + ToF rasterizer call (fwd/bwd), statistics, Adam" % (P, cfg["W"], cfg["H"]),
            "hip_ms": hip_ms, "hip_it_per_s": 1e3 / hip_ms,
            # the network's backward runs over the points with a non-zero upstream gradient row only (the Gaussians some
            # pixel blended); `network_backward_over_all_points_ms`: the same iteration with that switched off
            "network_backward_points": net_rows["points"], "network_backward_points_processed": net_rows["points_processed"],
            "network_backward_fraction": net_rows["points_processed"] / max(net_rows["points"], 1),
            "network_backward_over_all_points_ms": dense_ms,
            # the same iteration with the two rasterizer calls as one GaussianRasterizerPair (opt-in API)
            "hip_pair_ms": pair_ms, "hip_pair_it_per_s": 1e3 / pair_ms,
            "eager_glue_ms": eager_ms, "speedup": eager_ms / hip_ms}


def pair_extra(dev, scene, steps=20, warmup=5, which=("two", "pair")):
    """The colour-camera + ToF-camera calls of one iteration (gaussian_renderer/__init__.py:107-128) at the metric size:
    two GaussianRasterizer calls whose gradients autograd adds, against one GaussianRasterizerPair call (forward on two
    streams, one set of gradient tensors that the second view's backward adds its blended rows to).  Forward +
    backward of both views, the same two cameras a small baseline apart."""
    import numpy as np
    import torch
    from gftorf_amd import GaussianRasterizationSettings, GaussianRasterizer, GaussianRasterizerPair, synth
    cfg, g = scene["cfg"], scene["gaussians"]
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)

    def settings(cam, tof):
        return GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(scene["bg"]), scale_modifier=1.0,
            viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=cfg["D"], campos=t(cam["campos"]),
            prefiltered=False, debug=False, near_n=cam["znear"], far_n=cam["zfar"], depth_range=scene["depth_range"],
            use_view_dependent_phase=tof)
    cam_b = synth.make_camera(W, H, w2c=synth.look_at_w2c(0.0, 0.0, 0.0, (0.03, 0.0, 0.0)))     # ToF sensor beside the colour camera
    sa, sb = settings(scene["cam"], False), settings(cam_b, True)
    leaf = {k: t(v).requires_grad_(True) for k, v in g.items() if v is not None}
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
    gr = {k: t(v) for k, v in scene["grads"].items()}
    kw = dict(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
              scales=leaf["scales"], rotations=leaf["rotations"])
    ph, dc = scene["phase_offset"], scene["dc_offset"]
    ra, rb, rp = GaussianRasterizer(sa), GaussianRasterizer(sb), GaussianRasterizerPair(sa, sb)
    ups = [gr["color"], gr["phasor"], gr["depth"], gr["acc"], gr["depth_distortion"]]     # fixed upstream gradients, as in the headline step
    outs5 = lambda o: [o[0], o[1], o[2], o[4], o[6]]

    def clear():
        for v in leaf.values():
            v.grad = None
        m2.grad = None

    def two_calls():
        clear()
        torch.autograd.backward(outs5(ra(**kw)) + outs5(rb(phase_offset=ph, dc_offset=dc, **kw)), ups + ups)

    def pair():
        clear()
        oa, ob = rp(phase_offset=(0.0, ph), dc_offset=(0.0, dc), **kw)
        torch.autograd.backward(outs5(oa) + outs5(ob), ups + ups)

    def timed(fn):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            fn()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / steps

    if "two" not in which:                       # (profiling runs time one variant alone)
        return {"pair_ms": timed(pair)}
    two_ms = timed(two_calls)
    g_two = {k: v.grad.clone() for k, v in leaf.items()}
    if "pair" not in which:
        return {"two_calls_ms": two_ms}
    pair_ms = timed(pair)
    worst = max(float((v.grad - g_two[k]).abs().max() / (g_two[k].abs().max() + 1e-30)) for k, v in leaf.items())
    blended_b = None
    return {"what": "colour + ToF camera of one iteration, %d Gaussians, %dx%d, forward + backward of both views" % (P, W, H),
            "two_calls_ms": two_ms, "pair_ms": pair_ms, "speedup": two_ms / pair_ms, "pairs_per_s": 1e3 / pair_ms,
            "max_rel_gradient_difference": worst,
            "bytes_not_moved_per_pair": {"zero_gradient_rows_of_the_second_view": "376 B x (P - Gaussians the second view blended)",
                                         "autograd_sum_of_two_dense_gradient_sets": 3 * 376 * P}}


def views_extra(dev, scene, steps=90, warmup=30, views=30, moving_leg=True):
    """The headline step over VARYING views: 30 cameras on an arc around the same Gaussians in shuffled order (a training
    loop's access pattern), forward + backward each.  Only the size of the binning buffer comes from the previous
    frames of the shape -- other views here: restarted forwards (instance count above the guess), frames in which
    a quadrant walked past the sorted head of its list (that tile's list is completed on demand)."""
    import ctypes as C
    import random
    import numpy as np
    import torch
    from gftorf_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, api, synth
    cfg, g = scene["cfg"], scene["gaussians"]
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    bg = t(scene["bg"])
    rasts = []
    for v in range(views):
        a = (v / (views - 1) - 0.5) * 0.30
        cam = synth.make_camera(W, H, w2c=synth.look_at_w2c(yaw=a, pitch=0.04 * np.sin(3 * a), t=(-3.2 * np.sin(a), 0.0, 3.2 * (1 - np.cos(a)))))
        rasts.append(GaussianRasterizer(GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=bg, scale_modifier=1.0,
            viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=cfg["D"], campos=t(cam["campos"]),
            prefiltered=False, debug=False, near_n=cam["znear"], far_n=cam["zfar"], depth_range=scene["depth_range"],
            use_view_dependent_phase=scene["use_view_dependent_phase"])))
    leaf = {k: t(v).requires_grad_(True) for k, v in g.items() if v is not None}
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
    gr = {k: t(v) for k, v in scene["grads"].items()}
    ups = [gr["color"], gr["phasor"], gr["depth"], gr["acc"], gr["depth_distortion"]]
    rng = random.Random(7)
    order = []
    stats = dict(restarted=0)

    def step(record=False):
        if not order:
            order.extend(rng.sample(range(views), views))
        for x in leaf.values():
            x.grad = None
        m2.grad = None
        o = rasts[order.pop()](means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
                               scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=scene["phase_offset"],
                               dc_offset=scene["dc_offset"])
        torch.autograd.backward([o[0], o[1], o[2], o[4], o[6]], ups)
        if record:
            stats["restarted"] += int(api.last_call_stats["restarted"])

    for _ in range(warmup):
        step()
    legs = []
    for _leg in range(3):                    # three timed legs of `steps` frames: median + spread, as for the fog extra
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(True)
        torch.cuda.synchronize(dev)
        legs.append(time.perf_counter() - t0)
    legs.sort()
    dt = legs[1]
    # The same loop with the Gaussians MOVING between two visits of a camera, as in training (the legs above render static
    # Gaussians: every per-camera schedule -- list starts and capacities, tile hints, heavy-first order -- is exact there):
    # every frame the means drift by a seeded step of 0.2 % of the scene's depth range and opacities breathe by +-2 %; every
    # 30 frames (one round of the cameras) the opacities are reset to 0.05 for one round, the reference's opacity reset
    # (train.py:456-463, arguments/__init__.py:99).  Reported: it/s, schedule misses (frames binned twice), flagged quadrants.
    moving = None
    if moving_leg:
        gen = torch.Generator(device=dev).manual_seed(11)
        base_op = leaf["opacities"].detach().clone()
        misses0 = int(api.last_call_stats.get("sched_misses", 0))
        restarts0 = stats["restarted"]

        def perturb(i):
            with torch.no_grad():
                leaf["means3D"].add_(torch.randn(leaf["means3D"].shape, generator=gen, device=dev) * 0.01)
                if (i // views) % 4 == 3:
                    leaf["opacities"].fill_(0.05)
                else:
                    leaf["opacities"].copy_((base_op * (1.0 + 0.02 * torch.randn(base_op.shape, generator=gen, device=dev))).clamp_(0.005, 0.995))
        for i in range(2 * views):
            perturb(i)
            step()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        n_mov = 4 * views
        for i in range(n_mov):
            perturb(2 * views + i)
            step(True)
        torch.cuda.synchronize(dev)
        t_mov = time.perf_counter() - t0
        t0 = time.perf_counter()
        for i in range(n_mov):
            perturb(6 * views + i)
        torch.cuda.synchronize(dev)
        t_pert = time.perf_counter() - t0
        moving = {"what": "the same views with the Gaussians moving between two visits of a camera (drift of the means every frame, "
                          "an opacity reset to 0.05 for one round of the cameras in four)", "frames": n_mov,
                  "ms_per_step": (t_mov - t_pert) / n_mov * 1e3, "it_per_s": n_mov / max(t_mov - t_pert, 1e-9),
                  "ms_per_step_with_the_perturbation_itself": t_mov / n_mov * 1e3,
                  "list_schedule_misses": int(api.last_call_stats.get("sched_misses", 0)) - misses0,
                  "restarted_forwards": stats["restarted"] - restarts0}
        with torch.no_grad():
            leaf["opacities"].copy_(base_op)
    # per-stage HIP events over the same number of steps (second leg, as for the headline step)
    _lib.profile_reset()
    _lib.profile_enable(True)
    R_sum = walk_sum = flag_frames = flag_quads = app_sum = done_sum = 0
    for i in range(steps):
        step()
        if i % 10 == 0:                      # units of every tenth frame (each read is a device synchronisation)
            w = walked_instances(dev)
            R_sum += int(api.last_call_stats["num_rendered"])
            walk_sum += w["per_tile_deepest"] if w else 0
            flag_frames += int(bool(w and w["flagged_quadrants"]))
            flag_quads += w["flagged_quadrants"] if w else 0
            app_sum += (w["gaussians_with_appearance"] or 0) if w else 0
            done_sum += (w["completed_list_ids"] or 0) if w else 0
    torch.cuda.synchronize(dev)
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    n_units = (steps + 9) // 10
    calls = max(prof["forward_calls"], 1)
    stage_ms = {k[:-3]: prof[k] / calls for k in prof if k.endswith("_ms")}
    return {"what": "headline step over %d views on an arc in shuffled order, %d Gaussians, %dx%d, forward + backward" % (views, P, W, H),
            "it_per_s": steps / dt, "ms_per_step": dt / steps * 1e3, "steps": steps,
            "legs_ms_per_step": [l / steps * 1e3 for l in legs], "spread_of_legs": (legs[2] - legs[0]) / legs[1],
            "moving_gaussians": moving,
            "restarted_forwards": stats["restarted"],
            # frames whose lists did not fit the camera's list schedule (binned again by the counted flow) in this process so far
            "list_schedule_misses": int(api.last_call_stats.get("sched_misses", 0)),
            # of the sampled frames (every tenth): frames in which a quadrant walked past the sorted head of its list (the
            # silhouette quadrants of a view: their lists are completed on demand), and the means per sampled frame
            "sampled_frames": n_units, "sampled_frames_with_flagged_quadrants": flag_frames,
            "flagged_quadrants_per_frame": flag_quads / n_units, "completed_list_ids_per_frame": done_sum / n_units,
            "gaussians_with_appearance_per_frame": app_sum / n_units,
            "stage_ms": stage_ms, "gpu_ms_sum_of_stages": sum(stage_ms.values()),
            "roofline": views_roofline(stage_ms, P, W * H, ((W + 15) // 16) * ((H + 15) // 16), R_sum / n_units, walk_sum / n_units)}


def grads_kept_extra(dev, scene, steps=50, warmup=15):
    """The headline step for a caller that KEEPS its gradients (accumulation over several backwards,
    `zero_grad(set_to_none=False)`, an optimizer that holds `.grad`): the operator then cannot take back the gradient tensors
    of the previous backward and writes a fresh dense set every time -- 376 B of zeros per Gaussian nobody blended
    (GFT_GRADS_REUSE=0 gives the same path).  The other end of `gradient_tensors_reused` in the headline line."""
    import torch
    from gftorf_amd import _lib, api
    step, state, leaf = gpu_step_fn(scene, dev)
    sync = lambda: torch.cuda.synchronize(dev)
    keep = api._GRADS_REUSE
    api._GRADS_REUSE = False
    try:
        spin_up(step, sync)
        elapsed = timed_steps(step, steps, warmup, sync)
        reused = bool(api.last_call_stats.get("grads_reused"))
        _lib.profile_reset()
        _lib.profile_enable(True)
        for _ in range(steps):
            step()
        sync()
        prof = _lib.profile_read()
        _lib.profile_enable(False)
    finally:
        api._GRADS_REUSE = keep
    calls = max(prof["forward_calls"], 1)
    return {"what": "headline step with the gradient tensors written in full by every backward (the caller keeps them)",
            "it_per_s": steps / elapsed, "ms_per_step": elapsed / steps * 1e3, "gradient_tensors_reused": reused,
            "preprocess_bwd_ms": prof["preprocess_bwd_ms"] / calls}


def fog_extra(dev, steps=50, warmup=15):
    """The headline step on the `fog` workload (the metric frame with opacities 0.05-0.1: the regime the reference's own
    scenes start in, arguments/__init__.py:99): no pixel saturates, every list is walked whole, most visible Gaussians are
    blended -- what the lazy stages (list heads, appearance on demand, kept gradient tensors) cost or gain there."""
    import torch
    from gftorf_amd import _lib, api
    scene = build_scene("fog", 0, 1)
    step, state, leaf = gpu_step_fn(scene, dev)
    sync = lambda: torch.cuda.synchronize(dev)
    spin_up(step, sync)
    # three timed legs: the figure is their MEDIAN, the spread (max - min) / median stands beside it (a leg that starts on a
    # card some other extra has just left idle or hot reads differently: round 5's line said 1.61 ms where a run by itself
    # said 1.34)
    legs = sorted(timed_steps(step, steps, warmup if i == 0 else 3, sync) for i in range(3))
    elapsed = legs[1]
    _lib.profile_reset()
    _lib.profile_enable(True)
    for _ in range(steps):
        step()
    sync()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    cfg = scene["cfg"]
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
    P_vis = int((state["radii"] > 0).sum().item())
    P_blend = int((state["pixels"] > 0).sum().item())
    pairs = float(state["pixels"].double().sum().item())
    R = int(api.last_call_stats["num_rendered"])
    w = walked_instances(dev, state["radii"])
    units = {"P_app": w["gaussians_with_appearance"] or P_vis, "R_bin": (w["head_ids"] + w["completed_list_ids"]) if w["tile_pull"] else R,
             "R_walk": w["per_tile_deepest"], "P_blend": P_blend}
    if api.last_call_stats.get("grads_reused"):
        units.update(P_zero=0, P_rezero=P_blend)
    per_kernel, whole = algorithmic_bytes(P, P_vis, R, N, T, units=units)
    calls = max(prof["forward_calls"], 1)
    stage_ms = {k[:-3]: prof[k] / calls for k in prof if k.endswith("_ms")}
    dom = max(stage_ms, key=lambda k: stage_ms[k])
    ms = elapsed / steps * 1e3
    gbs = lambda b, m: b / (m * 1e-3) / 1e9 if m > 0 else 0.0
    cnt = load_counters(dom, "fog")
    return {"what": scene["label"], "it_per_s": steps / elapsed, "ms_per_step": ms, "steps": steps,
            "legs_ms_per_step": [l / steps * 1e3 for l in legs], "spread_of_legs": (legs[2] - legs[0]) / legs[1],
            "P_visible": P_vis, "gaussians_blended": P_blend, "blended_share_of_visible": P_blend / max(P_vis, 1),
            "num_rendered": R, "pair_evaluations": pairs, "flagged_quadrants": w["flagged_quadrants"],
            "stage_ms": stage_ms, "gpu_ms_sum_of_stages": sum(stage_ms.values()), "units_processed": units,
            "gradient_tensors_reused": bool(api.last_call_stats.get("grads_reused")),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": gbs(per_kernel[dom], stage_ms[dom]), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": gbs(per_kernel[dom], stage_ms[dom]) / HBM_PEAK_GBS,
                         "traffic": cnt["hbm_bytes"] if cnt else None,
                         "algorithmic_bytes_per_launch": per_kernel[dom], "avg_launch_ms": stage_ms[dom]},
            "path_roofline": {"algorithmic_bytes_per_step": whole, "frac": gbs(whole, ms) / HBM_PEAK_GBS}}


def c3_loop_extra(dev, iterations=7000):
    """BASELINE.json config 3 beside the headline: the torf `copier`-shaped optimisation loop of bench_loop.py (what
    `--workload C3 [--graph]` times on its own), all 7000 iterations, eagerly and with the iteration's device work replayed
    from a HIP graph per (SH degree, network on / off)."""
    import torch
    import bench_loop
    out = {"what": "C3: " + bench_loop.C3["label"] + "; %d iterations, it/s of the whole loop" % iterations}
    for key, graph in (("eager", False), ("graph", True)):
        def region(step_fn, n):
            for _ in range(20):
                step_fn()
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(n):
                step_fn()
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0
        secs, rep, info = bench_loop.run(dev, iterations, lambda: torch.cuda.synchronize(dev), region, graph=graph)
        out[key + "_it_per_s"] = iterations / secs
        out[key + "_ms_per_iteration"] = secs / iterations * 1e3
        out[key + "_loss_first_last"] = [rep["loss_trace"][0][1], rep["loss_trace"][-1][1]]
        del info
        torch.cuda.empty_cache()
    return out


def graph_pair_extra(dev, steps=200, warmup=30):
    """The two rasterizer calls of a C3-shaped iteration (100 k Gaussians at 320 x 240, opacity 0.1, two cameras on the same
    Gaussians) with their backward: eagerly through the blocking flow (the reference's call pattern: at this size the host --
    Python, autograd, the wait for the instance count -- bounds the loop, not the kernels), eagerly without the host read
    (`api.no_host_read`), and captured once in a torch.cuda.CUDAGraph and replayed (gft_forward_enqueue: no host read, so the
    iteration is capturable)."""
    import numpy as np
    import torch
    from gftorf_amd import GaussianRasterizationSettings, GaussianRasterizer, api, synth
    P, W, H = 100_000, 320, 240
    cfg = dict(P=P, W=W, H=H, D=3, sh_coeffs=16, tof=True, opacity_range=(0.1, 0.1))
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    scenes = [synth.make_scene(cfg, seed=77, w2c=synth.look_at_w2c(yaw=y, pitch=p_, t=(tx, 0.0, 0.0))) for y, p_, tx in ((0.03, -0.01, 0.02), (-0.05, 0.02, -0.04))]
    g = scenes[0]["gaussians"]
    leaf = {k: t(v).requires_grad_(True) for k, v in g.items() if v is not None}
    m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
    rasts, ups = [], []
    for sc in scenes:
        cam = sc["cam"]
        rasts.append(GaussianRasterizer(GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(sc["bg"]), scale_modifier=1.0,
            viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=3, campos=t(cam["campos"]), prefiltered=False,
            debug=False, near_n=cam["znear"], far_n=cam["zfar"], depth_range=sc["depth_range"], use_view_dependent_phase=True)))
        ups += [t(sc["grads"][k]) for k in ("color", "phasor", "depth", "acc", "depth_distortion")]
    sc0 = scenes[0]

    def iteration():
        for v in leaf.values():
            v.grad = None
        m2.grad = None
        outs = [r(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
                  scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=sc0["phase_offset"], dc_offset=sc0["dc_offset"]) for r in rasts]
        torch.autograd.backward([x for o in outs for x in (o[0], o[1], o[2], o[4], o[6])], ups)

    sync = lambda: torch.cuda.synchronize(dev)

    def timed(fn):
        for _ in range(warmup):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        return (time.perf_counter() - t0) / steps * 1e3
    eager_ms = timed(iteration)
    keep = api.no_host_read
    api.no_host_read = True
    try:
        nowait_ms = timed(iteration)
    finally:
        api.no_host_read = keep
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        iteration()
    torch.cuda.current_stream().wait_stream(side)
    sync()
    for v in leaf.values():
        v.grad = None
    m2.grad = None
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        iteration()
    replay_ms = timed(graph.replay)
    st = [x for x in api.enqueue_status() if x["key"][1:4] == (P, W, H)]
    return {"what": "two rasterizer calls + backward of a C3-shaped iteration (%d Gaussians, %dx%d, opacity 0.1), ms per iteration" % (P, W, H),
            "eager_blocking_ms": eager_ms, "eager_no_host_read_ms": nowait_ms, "graph_replay_ms": replay_ms,
            "speedup_replay_over_eager": eager_ms / replay_ms, "overflow": any(x["overflow"] for x in st)}


def views_roofline(stage_ms, P, N, T, R, R_walk):
    """`roofline` object of the varying-view step: dominant stage by HIP events, SURVEY 8(d) bytes x units processed
    (mean instance count / walked entries over the sampled frames; every visible Gaussian charged as P)."""
    dom = max(stage_ms, key=lambda k: stage_ms[k])
    per_kernel, whole = algorithmic_bytes(P, P, int(R), N, T, units={"R_walk": int(R_walk)})
    ms = stage_ms[dom]
    ach = per_kernel[dom] / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    return {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
            "traffic": None, "algorithmic_bytes_per_launch": per_kernel[dom], "avg_launch_ms": ms,
            "mean_instances": R, "mean_walked_entries": R_walk}


def knn_extra(dev, P=1_000_000):
    """SURVEY 8(f) row 3 beside the headline metric: distCUDA2 (mean squared distance to the 3
    nearest neighbours, the scale initialisation of scene/gaussian_model.py:194-199) on a
    clustered cloud; CPU baseline = the oracle's kd-tree port on a bounded sample."""
    import time
    import numpy as np
    import torch
    from gftorf_amd import distCUDA2
    from oracle import knn_ref
    rng = np.random.default_rng(5)
    c = rng.normal(0, 3, (8, 3))
    pts = (c[rng.integers(0, 8, P)] + rng.normal(0, 0.05, (P, 3)) + 20.0).astype(np.float32)
    pts[: P // 10] = rng.uniform(-10, 10, (P // 10, 3)).astype(np.float32) + 20.0
    t = torch.tensor(pts, device=dev)
    distCUDA2(t)
    torch.cuda.synchronize(dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        out = distCUDA2(t)
    b.record()
    torch.cuda.synchronize(dev)
    gpu_ms = a.elapsed_time(b) / 5
    n_cpu = 200_000
    t0 = time.time()
    ref = knn_ref.mean_dist2_kdtree(pts[:n_cpu])
    cpu_s = time.time() - t0
    sub = distCUDA2(t[:n_cpu]).cpu().numpy()
    err = float(np.max(np.abs(sub - ref) / np.maximum(ref, 1e-30)))
    return {"what": "distCUDA2, %d clustered points" % P, "gpu_ms": gpu_ms, "mpoints_per_s": P / gpu_ms / 1e3,
            "cpu_kdtree_port": {"points": n_cpu, "seconds": cpu_s, "mpoints_per_s": n_cpu / cpu_s / 1e6, "cores": 1},
            "max_rel_err_vs_oracle_on_sample": err, "checksum": float(out.double().sum().item())}


def adam_extra(dev, P=1_000_000, steps=10):
    """SURVEY 8(f) row 4 (optimizer part): one Adam step over the reference's per-Gaussian parameter
    groups (scene/gaussian_model.py:247-274: 91 floats per Gaussian), FusedAdam against
    torch.optim.Adam as the reference constructs it, same device.  28 B of HBM per element."""
    import torch
    from gftorf_amd import FusedAdam
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1, 1), (P, 15, 1), (P, 1, 1), (P, 15, 1), (P, 1), (P, 3), (P, 4), (P, 1, 3)]

    def make(cls):
        ps = [torch.nn.Parameter(torch.randn(s, device=dev)) for s in shapes]
        for p in ps:
            p.grad = torch.randn_like(p)
        return cls([{"params": [p], "lr": 1e-3} for p in ps], lr=0.0, eps=1e-15), ps

    def timed(cls):
        opt, ps = make(cls)
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize(dev)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            opt.step()
        b.record()
        torch.cuda.synchronize(dev)
        return a.elapsed_time(b) / steps, sum(p.numel() for p in ps)

    fused_ms, n = timed(FusedAdam)
    torch_ms, _ = timed(torch.optim.Adam)
    # opt-in step(visibility=...): 14 % of the Gaussians on screen, as in the metric frame (142 k of 1 M), in runs of
    # rows as a spatially sorted cloud has them
    vis = (torch.rand(P // 64 + 1, device=dev) < 0.14).repeat_interleave(64)[:P].contiguous()

    class Visible(FusedAdam):
        def step(self, closure=None):
            return super().step(closure, visibility=vis)
    vis_ms, _ = timed(Visible)
    nv = int(vis.sum()) * (n // P)
    return {"what": "Adam step, %d Gaussians x %d floats" % (P, n // P), "fused_ms": fused_ms, "torch_adam_ms": torch_ms,
            "speedup_vs_torch": torch_ms / fused_ms, "algorithmic_bytes": 28 * n,
            "achieved_GBs": 28 * n / (fused_ms * 1e-3) / 1e9, "frac_of_hbm_peak": 28 * n / (fused_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "visible_rows_only": {"what": "opt-in step(visibility=mask), not the reference's dense optimizer: %d of %d rows" % (int(vis.sum()), P),
                                  "ms": vis_ms, "algorithmic_bytes": 28 * nv + P, "achieved_GBs": (28 * nv + P) / (vis_ms * 1e-3) / 1e9}}


def spawn_ranks(n):
    """`bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py <same
    arguments>` as a child process (never an exec: this process stays the parent and has touched no GPU), pass the
    ranks' output through (rank 0 prints the JSON line) and return the launcher's exit code -- non-zero when any rank
    failed."""
    import socket
    import subprocess
    envc = dict(os.environ)
    envc.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    envc.setdefault("OMP_NUM_THREADS", "1")
    rc = 1
    for attempt in range(2):
        # (the port is free when asked and may be taken before the launcher binds it: one retry with another port)
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        r = subprocess.run(cmd, env=envc, stderr=subprocess.PIPE, text=True)
        sys.stderr.write(r.stderr)
        rc = r.returncode
        if rc == 0 or "Address already in use" not in r.stderr and "EADDRINUSE" not in r.stderr:
            break
    return rc


def main_cpu_rehearsal(args, env, world):
    """GFT_BENCH_REHEARSAL=cpu: the launch / rendezvous / timing plumbing of the N > 1 run without a GPU (gloo ranks, a
    host-side stand-in step, no rasterizer call): what tests/test_dist_gloo.py starts to check that `--gpus N` really
    runs N ranks.  The line it prints is marked as a rehearsal and carries no metric value."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group(backend="gloo")
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))
    frames = []

    def step():
        frames.append(frame_of_rank(env["rank"], len(frames), world))
        time.sleep(0.001)
    steps = min(args.steps, 20)
    elapsed = timed_steps(step, steps, min(args.warmup, 2), lambda: None, dist if world > 1 else None)
    if env["rank"] == 0:
        print(json.dumps({"rehearsal": "cpu: gloo ranks, host-side stand-in step, no rasterizer call", "metric": None, "value": None,
                          "n_gpus": world, "steps": steps, "ms_per_step": elapsed / steps * 1e3, "rccl_ranks": 0,
                          "process_group": ("gloo x%d" % dist.get_world_size()) if world > 1 else "none",
                          "frames_of_rank0": frames[-3:]}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 100, SURVEY 8(d); 7000 iterations for --workload C3)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="metric", choices=list(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-extras", action="store_true", help="skip the measurements beside the headline step")
    ap.add_argument("--extras", default="all", help="comma list of extras to run (render_pair, varying_views, c3_loop, assemble_inputs, knn, "
                                                    "adam, deform_network, densify, train_iteration) or `all`")
    ap.add_argument("--spin-up", type=float, default=0.3, help="seconds of untimed steps before the warm-up")
    ap.add_argument("--torch-loss", action="store_true", help="--workload C3: the loss's SSIM and L2 terms as stock torch (eight "
                    "convolutions + elementwise launches) instead of gftorf_amd.loss.ssim_l2")
    ap.add_argument("--graph", action="store_true", help="--workload C3: the iteration's device work captured in a HIP graph per "
                    "(SH degree, network on / off) and replayed (bench_loop.build_loop(graph=True)); eager is the default")
    ap.add_argument("--pair", action="store_true", help="--workload C3: the two rasterizer calls of an iteration as one "
                                                         "GaussianRasterizerPair (opt-in API; default: two calls, as the reference)")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 7000 if args.workload == "C3" else 100

    env = dist_env()
    world = max(env["world"], 1)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher (`python3 bench.py --gpus N`, the way the driver starts --gpus 1): this parent has
        # made no GPU call (torch is not even imported yet) and spawns the N ranks as fresh child processes
        sys.exit(spawn_ranks(args.gpus))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: one rank per GPU (start through torch.distributed.run, or "
                         "without WORLD_SIZE in the environment to let bench.py spawn the ranks)" % (args.gpus, world))
    rehearsal_mode = os.environ.get("GFT_BENCH_REHEARSAL", "")
    if rehearsal_mode == "cpu":
        return main_cpu_rehearsal(args, env, world)
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the rasterizer has no CPU path)")
    from gftorf_amd import _lib
    _lib.load()
    # GFT_BENCH_REHEARSAL=1: all ranks on device 0 with the gloo backend, to rehearse the N > 1 code path on
    # a one-GPU box (RCCL refuses two ranks on one device); never set by the driver
    rehearsal = rehearsal_mode == "1"
    local = 0 if rehearsal else env["local_rank"]
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist_mod.init_process_group(backend="gloo")
        else:
            dist_mod.init_process_group(backend="nccl", device_id=dev)
        dist = dist_mod
        if dist.get_world_size() != args.gpus:
            raise SystemExit("process group has %d ranks, --gpus %d" % (dist.get_world_size(), args.gpus))

    from gftorf_amd import api
    api.keep_last_buffers = True          # the walked-entries figure of path_roofline is read from the last forward's scratch
    if args.workload == "C3":
        return main_loop(args, env, world, dev, dist)
    scene = build_scene(args.workload, env["rank"], world)
    composed = bool(scene["cfg"].get("composed"))
    if composed:
        if dist is None:
            # one rank: the step still ends with its collective -- a single-rank RCCL group (the all-reduce is then a
            # device-side copy, but librccl is loaded, a communicator exists and its kernel is launched)
            try:
                import socket
                import torch.distributed as dist_mod
                with socket.socket() as so:
                    so.bind(("127.0.0.1", 0))
                    port = so.getsockname()[1]
                dist_mod.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                            device_id=dev)
                dist = dist_mod
            except Exception as e:
                print("bench.py: single-rank RCCL group unavailable (%s: %s): the step runs without its exchange"
                      % (type(e).__name__, e), file=sys.stderr)
        step, state, _ = c4_step_fn(scene, dev, dist, env["rank"], world)
    else:
        step, state, _ = gpu_step_fn(scene, dev)
    sync = torch.cuda.synchronize

    # Device spin-up before the W warm-up steps: the power management raises the clocks over the first
    # tens of milliseconds of load, and W steps of 0.65 ms are over before that (an occasional 30 % slower
    # timed leg with normal per-kernel times in the profiled leg was the symptom).  Untimed.
    t_spin = time.perf_counter()
    if composed and dist is not None and dist.get_world_size() > 1:
        # (every composed step holds a collective: all ranks must run the same number of them, so a count, not a time)
        for _ in range(max(1, int(args.spin_up / 0.003))):
            step()
    else:
        while time.perf_counter() - t_spin < args.spin_up:
            step()
        sync()
        # ... and until two consecutive chunks of 16 steps take the same time to 1 % (at most 1.5 s more): the ramp is not
        # equally long on every box / after every idle period (a default run read 0.4265 ms over its timed leg with a median of
        # 0.3884 in the leg behind it; 80 legs of 20 steps on another box: 0.387-0.390, one at 0.407)
        prev = None
        while time.perf_counter() - t_spin < args.spin_up + 1.5:
            t0 = time.perf_counter()
            for _ in range(16):
                step()
            sync()
            cur = time.perf_counter() - t0
            if prev is not None and abs(cur - prev) <= 0.01 * prev:
                break
            prev = cur
    sync()

    elapsed = timed_steps(step, args.steps, args.warmup, sync, dist)

    # ---- median leg ("median of 100"): the same K steps once more with one event per step on the launch stream;
    # `ms_per_step` stays the mean of the timed region above, `ms_per_step_median` is the median of these device-side step times
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    marks[0].record()
    for i in range(args.steps):
        step()
        marks[i + 1].record()
    sync()
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_ms = per_step[len(per_step) // 2] if per_step else None
    del marks

    # ---- roofline leg: the same K steps again with per-stage HIP events on the launch stream
    _lib.profile_reset()
    _lib.profile_enable(True)
    for _ in range(args.steps):
        step()
    sync()
    prof = _lib.profile_read()
    _lib.profile_enable(False)

    cfg = scene["cfg"]
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    N = W * H
    T = ((W + 15) // 16) * ((H + 15) // 16)
    radii = state["radii"]
    P_vis = int((radii > 0).sum().item())
    R = int(api.last_call_stats["num_rendered"])
    walked = walked_instances(dev, radii)
    pairs = float(state["pixels"].double().sum().item())      # (pixel, Gaussian) pairs that were blended
    P_blend = int((state["pixels"] > 0).sum().item())         # Gaussians with a non-zero gradient row
    restarts = (api.last_call_stats.get("restarts", 0), api.last_call_stats.get("forwards", 0))

    exchange = None
    composed_step = None
    if composed:
        fs = state["frame_step"]
        # phases of the composed step by events on the launch stream (a third pass of 20 steps)
        marks = []
        fs.mark = lambda name: marks.append((name, torch.cuda.Event(enable_timing=True))) or marks[-1][1].record()
        for _ in range(20):
            step()
        sync()
        fs.mark = None
        phase = {}
        for (n0, e0), (n1, e1) in zip(marks, marks[1:]):
            if n1 != "start":
                phase[n1] = phase.get(n1, 0.0) + e0.elapsed_time(e1) / 20
        n_dyn = int(fs.x_norm.size(0))
        n_in = fs.net.xyz_input_ch + fs.net.t_input_ch
        macs_fwd = n_in * 256 + 6 * 256 * 256 + (n_in + 256) * 256 + 51 * 256
        tf_fwd = 2.0 * macs_fwd * n_dyn / (phase["network_forward"] * 1e-3) / 1e12 if phase.get("network_forward") else 0.0
        fwd_terms = 3.0 if os.environ.get("GFT_DEFORM_FP16X2", "1") != "0" else 6.0
        composed_step = {"phase_ms": phase, "gpu_ms_sum_of_phases": sum(phase.values()),
                         # the largest kernel of the composed step is the deformation network's forward walk (all dynamic
                         # points: positions decide visibility, it cannot be thinned); fp32 results from three fp16 MFMAs per
                         # product (six bf16 ones with GFT_DEFORM_FP16X2=0)
                         "dominant": {"kernel": "k_deform_fwd_h" if fwd_terms == 3.0 else "k_deform_fwd_bf", "bound": "mfma", "achieved": tf_fwd,
                                      "peak": BF16_MFMA_PEAK_TFLOPS / fwd_terms,
                                      "unit": "TFLOP/s", "frac": tf_fwd / (BF16_MFMA_PEAK_TFLOPS / fwd_terms), "points": n_dyn,
                                      "avg_launch_ms": phase.get("network_forward"),
                                      "note": "peak = 2500 TFLOP/s of fp16 MFMA at 2.4 GHz over the multiplies per fp32 product; the "
                                              "kernel runs at the board's power limit (1372-1376 W of 1400, engine clock 1.5-2.1 GHz "
                                              "while it runs: profiles/r04/r04_power_clock_samples.txt, DESIGN 10.1) -- the gap to the peak "
                                              "is energy per point, not issue slots; avg_launch_ms is the phase between two events "
                                              "(the walk plus the three pack kernels of the call)"}}
        exchange = {"in_the_timed_step": True, "collectives_per_step": fs.exchanges / max(state["it"], 1),
                    "bucket_bytes": fs.exchanged_bytes, "backend": dist.get_backend() if dist is not None else None,
                    "ranks": dist.get_world_size() if dist is not None else 0,
                    "dynamic_gaussians": int(fs.x_norm.size(0)), "frames": fs.num_frames}
    elif dist is not None and not args.no_extras:
        # the deform-network exchange leg runs on every rank (it holds the path's only collective)
        try:
            exchange = deform_exchange(dev, dist, env["rank"], world)
        except Exception as e:                            # reported, never fatal for the headline line
            exchange = {"error": "%s: %s" % (type(e).__name__, e)}

    my_calls = max(prof["forward_calls"], 1)
    my_stage_ms = {k[:-3]: prof[k] / my_calls for k in prof if k.endswith("_ms")}
    per_rank_stage_ms = None
    if dist is not None and dist.get_world_size() > 1:
        per_rank_stage_ms = [None] * dist.get_world_size()
        dist.all_gather_object(per_rank_stage_ms, my_stage_ms)

    if env["rank"] == 0:
        fo = bool(cfg.get("forward_only"))
        # SURVEY 8(d) as written (reference algorithm) and the same constants on the units the launches process
        ref_kernel, ref_whole = algorithmic_bytes(P, P_vis, R, N, T, forward_only=fo)
        units = None
        if walked:
            # tile-pull binning: the supertile entries are what is counted and scattered, the list heads (+ completed lists)
            # what is sorted; R_bin charges the larger of the two per instance-equivalent
            units = {"P_app": walked["gaussians_with_appearance"],
                     "R_bin": (walked["head_ids"] + walked["completed_list_ids"]) if walked["tile_pull"] else R,
                     "R_walk": walked["per_tile_deepest"], "P_blend": P_vis if fo else P_blend}
            if not fo and api.last_call_stats.get("grads_reused"):
                # the gradient tensors were kept from the previous backward: its rows zeroed again, only this one's written
                units.update(P_zero=0, P_rezero=P_blend)
        per_kernel, whole = algorithmic_bytes(P, P_vis, R, N, T, forward_only=fo, units=units)
        calls = max(prof["forward_calls"], 1)
        stage_ms = {k[:-3]: prof[k] / calls for k in prof if k.endswith("_ms")}
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        dom_ms = stage_ms[dom]
        gbs = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        achieved = gbs(per_kernel[dom], dom_ms)
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        cnt = load_counters(dom, args.workload)
        traffic = cnt["hbm_bytes"] if cnt else None
        all_cnt = [load_counters(k, args.workload) for k in stage_ms if k != "memset"]
        path_traffic = sum(c["hbm_bytes"] for c in all_cnt) if all(all_cnt) else None
        # `achieved` / `frac`: SURVEY 8(d)'s per-unit bytes x the units this launch processes (for the render stages the
        # list entries walked) / the kernel's mean duration by HIP events.  `traffic`: HBM bytes of one launch by the
        # committed --pmc passes.  `reference_formula`: the same per-unit bytes x every instance, as the reference
        # algorithm moves them -- kept for comparison with round 1; it is not a bound for a lazy implementation.
        roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": per_kernel[dom], "avg_launch_ms": dom_ms,
                    "units_processed": units,
                    "counter_frac": (gbs(traffic, dom_ms) / HBM_PEAK_GBS) if (traffic and dom_ms > 0) else None,
                    # traffic = factor x FETCH_SIZE + WRITE_SIZE, the factor by the kernel's access kind as calibrated on the box
                    # (profiles/fetch_factors.py, profiles/r06_fetch_calibration.json): 2 for wide streams, 1 for record gathers
                    "traffic_fetch_factor": cnt.get("fetch_factors") if cnt else None,
                    "counter_source": cnt["source"] if cnt else None,
                    # the counter passes were taken from these kernel sources / the sources that ran now
                    "counter_source_sha": cnt.get("kernel_source_sha") if cnt else None, "kernel_source_sha": kernel_source_sha(),
                    "counters_belong_to_this_source": bool(cnt) and cnt.get("kernel_source_sha") == kernel_source_sha(),
                    "reference_formula": {"algorithmic_bytes_per_launch": ref_kernel[dom], "achieved": gbs(ref_kernel[dom], dom_ms),
                                          "frac": gbs(ref_kernel[dom], dom_ms) / HBM_PEAK_GBS,
                                          "note": "SURVEY 8(d) x all instances (what the reference algorithm moves); early "
                                                  "termination makes most of it dead data: not a bound"}}
        if dom in FLOPS_PER_PAIR:
            # the render kernels are bound by VALU issue, not by HBM: report that ceiling beside the byte figures
            useful = pairs * FLOPS_PER_PAIR[dom]
            roofline["valu"] = {
                "pair_evaluations": pairs, "flops_per_pair": FLOPS_PER_PAIR[dom],
                "useful_TFLOPs": useful / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else None,
                "peak_TFLOPs": FP32_VALU_PEAK_TFLOPS,
                "useful_frac_of_fp32_valu_peak": useful / (dom_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS if dom_ms > 0 else None,
                "wave_instructions_per_launch": cnt.get("valu_insts") if cnt else None,
                "resident_waves_per_simd": cnt.get("mean_resident_waves_per_simd") if cnt else None,
                "issue_slot_frac": cnt.get("valu_issue_slot_frac") if cnt else None,
                "source": cnt["source"] if cnt else None}
        out = {
            "metric": "train iters/sec (fwd+bwd raster) + Mpix/s, 1M Gaussians @ 640x480 ToF",
            "value": value, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "ms_per_step_median": median_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            # ranks of the process group the timed region's barriers ran on (backend nccl = RCCL; gloo in a rehearsal)
            "rccl_ranks": dist.get_world_size() if (dist is not None and dist.get_backend() == "nccl") else (1 if dist is None else 0),
            "process_group": ("%s x%d" % (dist.get_backend(), dist.get_world_size())) if dist is not None else "none",
            "config": {"workload": scene["label"], "P": P, "W": W, "H": H, "sh_degree": cfg["D"],
                       "P_visible": P_vis, "num_rendered": R,
                       "longest_tile_list": int(api.last_call_stats["max_tile_list"]), "frames_per_step": world,
                       "parallelism": ("frame-sharded x%d (no collective in the raster path; one all-reduce of the deform-network "
                                       "gradient bucket per step, inside the timed step: deform_exchange)" % world) if composed else
                                      ("frame-sharded x%d (no collective in the raster path; the deform-gradient all-reduce is "
                                       "measured in deform_exchange)" % world)},
            "mpix_per_s": value * N / 1e6,
            "roofline": roofline,
            # whole step: per-unit bytes of SURVEY 8(d) x units processed (P_app, R_bin, R_walk, P_blend, see
            # algorithmic_bytes) / step time; `reference_formula_*`: 8(d) as written for the reference algorithm.
            # 6290 GB/s is the measured copy ceiling of the part.
            "path_roofline": {"algorithmic_bytes_per_step": whole,
                              "achieved_GBs": gbs(whole, ms_per_step),
                              "frac": gbs(whole, ms_per_step) / HBM_PEAK_GBS,
                              "per_stage_bytes": per_kernel,
                              "instances": R, "instances_binned": units["R_bin"] if units else R,
                              "instances_walked": units["R_walk"] if units else R,
                              "instances_walked_per_quadrant_sum": walked["per_quadrant_sum"] if walked else None,
                              "gaussians_visible": P_vis, "gaussians_with_appearance": (units or {}).get("P_app") or P_vis,
                              "gaussians_blended": P_blend if not fo else None,
                              "reference_formula_bytes": ref_whole,
                              "reference_formula_frac": gbs(ref_whole, ms_per_step) / HBM_PEAK_GBS,
                              "achievable_copy_GBs": 6290.0,
                              # HBM bytes of all stages as the counters saw them (profiles/counters.json)
                              "counter_bytes_per_step": path_traffic,
                              "counter_frac": (gbs(path_traffic, ms_per_step) / HBM_PEAK_GBS) if path_traffic else None,
                              "gpu_ms_sum_of_stages": sum(stage_ms.values())},
            "binning_restarts": {"restarted_forwards": restarts[0], "forwards": restarts[1],
                                 "list_schedule_misses": int(api.last_call_stats.get("sched_misses", 0))},
            # gradient tensors kept from one backward to the next (their written rows re-zeroed) instead of 376 B of zeros
            # per Gaussian written in every backward; only when no tensor aliases them any more and nobody wrote to them
            "gradient_tensors_reused": bool(api.last_call_stats.get("grads_reused")),
            "tile_pull_binning": {k: walked[k] for k in ("tile_pull", "supertile_entries", "head_ids", "flagged_quadrants",
                                                         "completed_list_ids", "gaussians_with_appearance")} if walked else None,
            "stage_ms": stage_ms,
        }
        if per_rank_stage_ms is not None:
            out["per_rank_stage_ms"] = per_rank_stage_ms
        if exchange is not None:
            out["deform_exchange"] = exchange
            # (the N > 1 line's exchange figures where a reader of the top level finds them)
            for k in ("allreduce_alone_ms", "replicas_identical"):
                if k in exchange:
                    out[k] = exchange[k]
        if composed_step is not None:
            out["composed_step"] = composed_step
            # the composed step's dominant kernel is the network's forward walk (matrix-core bound), not a rasterizer kernel:
            # `roofline` describes it; the rasterizer's own dominant kernel moves to `raster_roofline`
            out["raster_roofline"] = out["roofline"]
            out["roofline"] = dict(composed_step["dominant"], traffic=None)
            out["metric"] = ("composed frame steps/sec: network query + input assembly + raster fwd+bwd + network backward + "
                             "gradient all-reduce, 1M Gaussians (30 % dynamic) @ 640x480 ToF, one frame per rank")
        # (the measurements beside the headline belong to the metric workload; named explicitly they run on any workload's
        # scene, e.g. `--workload C5 --extras varying_views`)
        if world == 1 and not args.no_extras and (args.workload == "metric" or args.extras != "all"):
            del state, step
            torch.cuda.empty_cache()
            table = [("train_iteration", lambda: train_iteration_extra(dev, scene)), ("render_pair", lambda: pair_extra(dev, scene)),
                     ("varying_views", lambda: views_extra(dev, scene)), ("fog", lambda: fog_extra(dev)), ("graph_pair", lambda: graph_pair_extra(dev)), ("c3_loop", lambda: c3_loop_extra(dev)), ("grads_kept", lambda: grads_kept_extra(dev, scene)), ("assemble_inputs", lambda: assemble_extra(dev)),
                     ("knn", lambda: knn_extra(dev)), ("adam", lambda: adam_extra(dev)), ("deform_network", lambda: deform_extra(dev)),
                     ("densify", lambda: densify_extra(dev))]
            want = None if args.extras == "all" else set(args.extras.split(","))
            out["extras"] = {}
            for name, fn in table:
                if want is None or name in want:
                    # (a measurement beside the headline that fails says so in its place: the headline line still prints)
                    try:
                        out["extras"][name] = fn()
                    except Exception as e:
                        print("bench.py: extra `%s` failed: %s: %s" % (name, type(e).__name__, e), file=sys.stderr)
                        out.setdefault("extras_failed", {})[name] = "%s: %s" % (type(e).__name__, str(e)[:300])
                    torch.cuda.empty_cache()
            # the training-representative figures beside the repeated-frame headline, at the top level of the line: shuffled
            # views (every frame another camera), a caller that keeps its gradients (dense gradient writes), the frame in
            # which nothing saturates (every list walked whole) and that frame's share of the HBM roofline by units processed
            ex = out["extras"]
            if "varying_views" in ex:
                out["varying_views_it_per_s"] = ex["varying_views"]["it_per_s"]          # (median of three legs)
                out["varying_views_spread"] = ex["varying_views"]["spread_of_legs"]
                if ex["varying_views"].get("moving_gaussians"):
                    mv = ex["varying_views"]["moving_gaussians"]
                    out["moving_gaussians_it_per_s"] = mv["it_per_s"]
                    out["moving_gaussians_schedule_misses"] = "%d of %d frames" % (mv["list_schedule_misses"], mv["frames"])
            if "grads_kept" in ex:
                out["grads_kept_it_per_s"] = ex["grads_kept"]["it_per_s"]
            if "graph_pair" in ex:
                out["c3_pair_eager_ms"] = ex["graph_pair"]["eager_blocking_ms"]
                out["c3_pair_graph_replay_ms"] = ex["graph_pair"]["graph_replay_ms"]
            if "c3_loop" in ex:
                out["c3_loop_it_per_s"] = ex["c3_loop"]["eager_it_per_s"]                # BASELINE config 3 (bench.py --workload C3)
                out["c3_loop_graph_it_per_s"] = ex["c3_loop"]["graph_it_per_s"]          # ... --graph
            if "fog" in ex:
                out["fog_it_per_s"] = ex["fog"]["it_per_s"]
                out["fog_ms_per_step"] = ex["fog"]["ms_per_step"]                        # (median of three legs)
                out["fog_spread"] = ex["fog"]["spread_of_legs"]
                out["fog_path_frac"] = ex["fog"]["path_roofline"]["frac"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, budget_s=args.cpu_budget, forward_only=fo)
            out["speedup_vs_cpu"] = value / out["cpu_baseline"]["value"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main_loop(args, env, world, dev, dist):
    """--workload C3: BASELINE.json config 3, the torf `copier`-shaped optimisation loop (bench_loop.py).  A step is one
    training iteration; every rank optimises its own replica of the scene (no collective: replicas only)."""
    import torch
    import bench_loop
    from gftorf_amd import _lib, api
    sync = torch.cuda.synchronize
    holder = {}

    def region(step_fn, n):
        holder["step"] = step_fn
        return timed_steps(step_fn, n, args.warmup, sync, dist)
    elapsed, rep, info = bench_loop.run(dev, args.steps, sync, region, pair=args.pair, graph=args.graph, fused_loss=not args.torch_loss)
    # roofline leg: per-stage HIP events of the rasterizer calls over 200 more iterations
    _lib.profile_reset()
    _lib.profile_enable(True)
    n_prof = 200
    for _ in range(n_prof):
        holder["step"]()
    sync()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    if env["rank"] == 0:
        cfg = bench_loop.C3
        P, W, H = cfg["P"], cfg["W"], cfg["H"]
        N, T = W * H, ((W + 15) // 16) * ((H + 15) // 16)
        R = int(api.last_call_stats["num_rendered"])
        walked = walked_instances(dev)
        fwd_calls, bwd_calls = max(prof["forward_calls"], 1), max(prof["backward_calls"], 1)
        # two forwards and one backward per iteration: per-call stage times
        stage_ms = {k[:-3]: prof[k] / (bwd_calls if k in ("render_bwd_ms", "preprocess_bwd_ms", "memset_ms") else fwd_calls)
                    for k in prof if k.endswith("_ms")}
        raster_ms_per_it = sum(prof[k] for k in prof if k.endswith("_ms")) / n_prof
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        per_kernel, whole = algorithmic_bytes(P, P, R, N, T, units={"R_walk": walked["per_tile_deepest"]} if walked else None)
        ms = elapsed / args.steps * 1e3
        value = world * args.steps / elapsed
        achieved = per_kernel[dom] / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        out = {
            "metric": "train iters/sec (fwd+bwd raster) + Mpix/s, 1M Gaussians @ 640x480 ToF",
            "value": value, "unit": "it/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C3: " + cfg["label"], "P": P, "W": W, "H": H, "views": cfg["views"], "sh_degree": cfg["sh_degree"],
                       "warm_up": cfg["warm_up"], "num_rendered_last_call": R,
                       "step": "one training iteration: LR schedule, random view + background, deform query (after warm-up), "
                               "activations, input assembly, colour + ToF rasterizer forward, ToF loss (L2 + SSIM), backward, "
                               "densification statistics, Adam (Gaussians + network)",
                       "rasterizer_calls": "GaussianRasterizerPair" if args.pair else "two GaussianRasterizer calls",
                       "iteration_runs_as": "HIP graph replay per (SH degree, network on / off), captured after 3 eager iterations of "
                                            "the configuration" if args.graph else "eager launches",
                       "parallelism": "replicas x%d" % world},
            "mpix_per_s": value * 2 * N / 1e6,
            "loop": rep,
            "raster_ms_per_iteration": raster_ms_per_it, "raster_share_of_iteration": raster_ms_per_it / ms,
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": per_kernel[dom],
                         "avg_launch_ms": stage_ms[dom],
                         "note": "at 100k Gaussians / 320x240 a rasterizer call is ~0.1 ms of kernels: the iteration is bound by "
                                 "launch latency and the host, not by HBM (raster_share_of_iteration)"},
            "stage_ms": stage_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            # CPU beside it: the oracle's rasterizer forward+backward on one C3-shaped ToF frame (the rest of the loop is
            # host-side torch in the reference as well)
            from gftorf_amd import synth
            g0, cams = info["frame"]["g0"], info["frame"]["cams"]
            cam = {k: v for k, v in cams[len(cams) // 2][1].items() if not k.startswith("_")}
            frame = dict(cfg=dict(P=P, W=W, H=H, D=3, sh_coeffs=16, tof=True), cam=cam, gaussians=g0,
                         bg=synth.make_background(W, H, 3), grads=synth.make_pixel_grads(W, H, 3), depth_range=cfg["depth_range"],
                         phase_offset=0.1, dc_offset=0.0, use_view_dependent_phase=True)
            cb = cpu_baseline(frame, budget_s=args.cpu_budget)
            cb["sample"] = "rasterizer forward+backward of one C3 ToF frame (100k Gaussians, 320x240): " + cb["sample"]
            cb["unit"] = "raster fwd+bwd/s"
            out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
