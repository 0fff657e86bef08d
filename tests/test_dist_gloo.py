"""N > 1 path of bench.py on CPU: two gloo ranks, frame sharding without a data-path
collective, barrier-bracketed timing with MAX over ranks."""
import os
import sys
import time

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    """A port the OS says is free right now (a fixed one derived from the pid met a port in use once in a while)."""
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import bench
    dist.init_process_group("gloo", rank=rank, world_size=world)
    env = bench.dist_env()
    assert env == dict(rank=rank, local_rank=rank, world=world)
    frames = [bench.frame_of_rank(rank, s, world) for s in range(3)]
    # rank 1 is slower: the reported time must be the slowest rank's
    delay = 0.02 * (rank + 1)
    calls = []

    def step():
        calls.append(1)
        time.sleep(delay)

    dt = bench.timed_steps(step, steps=5, warmup=2, sync_fn=lambda: None, dist=dist)
    q.put((rank, frames, len(calls), dt))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_frame_sharding_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, f0, c0, t0), (r1, f1, c1, t1) = res
    # disjoint frames that tile the frame index space
    assert f0 == [0, 2, 4] and f1 == [1, 3, 5]
    assert c0 == c1 == 7                       # 2 warm-up + 5 timed
    assert abs(t0 - t1) < 1e-9                 # both ranks report the MAX
    assert t0 >= 5 * 0.04 * 0.9                # ... which is the slow rank's time


def _worker_deform(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from gftorf_amd.deform import reference_network, allreduce_gradients, flat_grad_bucket
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                       # replicas start identical
    net = reference_network()
    # the gradients a rank's own frame produced (stand-ins: the kernels need a GPU); rot / a get none
    g = torch.Generator().manual_seed(100 + rank)
    for name, p in net.named_parameters():
        if not name.startswith(("rot.", "a.")):
            p.grad = torch.randn(p.shape, generator=g)
    mine = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    nbytes = allreduce_gradients(net, dist, average=True)
    flat, _ = flat_grad_bucket(net)
    q.put((rank, nbytes, {n: v for n, v in mine.items() if n in ("linear.5.weight", "b.bias")},
           {n: p.grad.clone() for n, p in net.named_parameters() if n in ("linear.5.weight", "b.bias")},
           float(flat.double().sum()), [n for n, p in net.named_parameters() if p.grad is None]))
    dist.barrier()
    dist.destroy_process_group()


def _worker_deform_views(rank, world, port, q):
    """As _worker_deform, with the gradients laid out as the HIP backward leaves them: consecutive views of ONE buffer, each
    on a 16-byte boundary -- flat_grad_bucket must hand that memory to the all-reduce itself."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from gftorf_amd.deform import reference_network, allreduce_gradients, flat_grad_bucket, _param_list
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    net = reference_network()
    ps = _param_list(net)
    sizes = [p.numel() for p in ps]
    buf = torch.zeros(sum((k + 3) // 4 * 4 for k in sizes))
    g = torch.Generator().manual_seed(200 + rank)
    o = 0
    for p, k in zip(ps, sizes):
        p.grad = buf[o:o + k].view(p.shape)
        p.grad.copy_(torch.randn(p.shape, generator=g))
        o += (k + 3) // 4 * 4
    mine = buf.clone()
    flat, _ = flat_grad_bucket(net)
    in_place = flat.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr() and flat.numel() == buf.numel() - (buf.numel() - o) - ((sizes[-1] + 3) // 4 * 4 - sizes[-1])
    nbytes = allreduce_gradients(net, dist, average=True)
    q.put((rank, nbytes, bool(in_place), mine.numpy(), buf.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allreduce_in_the_gradients_own_buffer():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_deform_views, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, nb0, ip0, mine0, red0), (_, nb1, ip1, mine1, red1) = res
    assert ip0 and ip1                                        # no gathered copy was made
    assert nb0 == nb1 == (522055 - 5140 + 1) * 4              # the used parameters + one padding float behind xyz_warp.bias
    import numpy as np
    np.testing.assert_allclose(red0, (mine0 + mine1) / 2, rtol=1e-6, atol=1e-7)
    assert (red0 == red1).all()                               # p.grad of both replicas IS the reduced buffer


def test_two_rank_deform_gradient_allreduce():
    """The one exchange step of the path: all-reduce of the deformation network's gradient
    bucket; here over gloo with two CPU ranks."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_deform, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, nb0, mine0, red0, sum0, none0), (_, nb1, mine1, red1, sum1, none1) = res
    assert nb0 == nb1 == (522055 - 5140) * 4                # every parameter that receives a gradient, once
    assert none0 == none1 == ["rot.weight", "rot.bias", "a.weight", "a.bias"]
    for n in ("linear.5.weight", "b.bias"):
        torch.testing.assert_close(red0[n], (mine0[n] + mine1[n]) / 2)
        assert torch.equal(red0[n], red1[n])                  # replicas hold identical gradients afterwards
    assert sum0 == sum1


def test_algorithmic_bytes_formula():
    sys.path.insert(0, ROOT)
    import bench
    per, whole = bench.algorithmic_bytes(P=1_000_000, P_vis=910_000, R=3_000_000, N=640 * 480, T=1200)
    # 1516*P_vis + 432*P_cull + 268*R + 224*N ~ 2.3 GB
    assert whole == 1516 * 910_000 + 432 * 90_000 + 268 * 3_000_000 + 224 * 307_200
    assert abs(whole / 1e9 - 2.29) < 0.05
    # the per-kernel split adds up to the whole-path figure
    assert sum(per.values()) == whole
    per_f, whole_f = bench.algorithmic_bytes(P=1_000_000, P_vis=910_000, R=3_000_000, N=640 * 480, T=1200, forward_only=True)
    assert whole_f == 512 * 910_000 + 48 * 90_000 + 120 * 3_000_000 + 128 * 307_200
    # charged for the units a lazy implementation processes: the same constants, never more than the formula as written,
    # equal to it when every unit is processed
    full = dict(P_app=910_000, R_bin=3_000_000, R_walk=3_000_000, P_blend=910_000)
    assert bench.algorithmic_bytes(1_000_000, 910_000, 3_000_000, 307_200, 1200, units=full)[1] == whole
    lazy = dict(P_app=300_000, R_bin=1_100_000, R_walk=600_000, P_blend=250_000)
    per_l, whole_l = bench.algorithmic_bytes(1_000_000, 910_000, 3_000_000, 307_200, 1200, units=lazy)
    assert whole_l < whole and all(per_l[k] <= per[k] for k in per)
    assert per_l["render_bwd"] == 148 * 600_000 + 96 * 307_200
    assert per_l["preprocess_bwd"] == 928 * 250_000 + 384 * 750_000


def test_bare_bench_gpus2_spawns_two_ranks():
    """`python3 bench.py --gpus 2` started the way the driver starts `--gpus 1` (no launcher, no WORLD_SIZE): the
    parent spawns 2 ranks as child processes and relays rank 0's line, which says n_gpus 2 -- it must never run one
    rank and call it two.  GFT_BENCH_REHEARSAL=cpu: gloo ranks and a host-side stand-in step (no GPU here)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GFT_BENCH_REHEARSAL"] = "cpu"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["process_group"] == "gloo x2" and lines[0]["steps"] == 5
    # a world size that contradicts --gpus is refused, not silently accepted
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env2, capture_output=True, text=True, timeout=120)
    assert r2.returncode != 0 and "WORLD_SIZE=1" in (r2.stdout + r2.stderr)


def _frame_step_parts(P=500, W=64, H=48):
    """A small scene and the host-side stand-ins of the composed C4 step: the CPU oracle behind the operator's autograd
    surface, the eager input assembly, the reference network's eager statements on this package's module (same
    parameters, names and order)."""
    import numpy as np
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers as Hh
    from oracle import assemble_ref, deform_ref, oracle
    from gftorf_amd.deform import DeformNetwork, REFERENCE_ARCH
    oracle.build()

    class EagerNet(DeformNetwork):
        def forward(self, x, t):
            return deform_ref.deform_eager(dict(self.named_parameters()), x, t)

    scene = Hh.small_scene(P=P, W=W, H=H, seed=17)
    g = scene["gaussians"]
    torch.manual_seed(3)                                   # replicas start identical
    net = EagerNet(**REFERENCE_ARCH)
    for name, p in net.named_parameters():
        torch.nn.init.normal_(p, 0.0, 0.05 if (name.startswith("linear") and name.endswith("weight")) else 2e-3)
    t = lambda a: torch.tensor(np.asarray(a, np.float32))
    leaf = dict(xyz=t(g["means3D"]), opacity=t(g["opacities"]).reshape(P, 1), scaling=t(g["scales"]),
                rotation_raw=t(g["rotations"]), fc=t(g["shs"]), fp=t(g["shs_p"]))
    for v in leaf.values():
        v.requires_grad_(True)
    mask = torch.tensor(np.random.default_rng(4).random(P) < 0.4)
    upstream = [t(scene["grads"][k]) for k in Hh.GRAD_KEYS]
    return scene, net, leaf, mask, upstream, Hh.oracle_rasterizer(oracle, scene), assemble_ref.assemble_eager


def _worker_frame_step(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    import bench
    from gftorf_amd.frames import FrameStep
    from gftorf_amd.deform import flat_grad_bucket
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene, net, leaf, mask, upstream, render, assemble = _frame_step_parts()
    calls = []

    def exchange(module, d, average=True):
        from gftorf_amd.deform import allreduce_gradients
        calls.append(1)
        return allreduce_gradients(module, d, average=average)
    fs = FrameStep(net, leaf, mask, render, upstream, dist=dist, num_frames=8, assemble=assemble, exchange=exchange)
    local = FrameStep(net, leaf, mask, render, upstream, dist=None, num_frames=8, assemble=assemble)
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, eps=1e-15)
    mine, reduced, frames = None, None, []
    for it in range(3):
        frame = bench.frame_of_rank(rank, it, world)
        frames.append(frame)
        if it == 0:
            local.zero_grad()
            local(frame)                                   # this rank's own gradients, no exchange
            mine = flat_grad_bucket(net)[0].clone()
        fs.zero_grad()
        fs(frame)
        if it == 0:
            reduced = flat_grad_bucket(net)[0].clone()
        opt.step()                                         # identical gradients -> identical replicas
    weights = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    # (numpy: pickled by value -- a tensor travels as a file descriptor that dies with this process)
    q.put((rank, frames, len(calls), fs.exchanges, fs.exchanged_bytes, mine.numpy(), reduced.numpy(), weights.numpy(),
           float(leaf["xyz"].grad.abs().sum()), fs.frame_time(frames[0])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_composed_frame_step():
    """BASELINE.json config 4's step (gftorf_amd.frames.FrameStep: deform query at the rank's frame time -> input assembly
    -> rasterizer forward + backward -> network backward -> gradient all-reduce), rehearsed by two gloo ranks with
    host-side stand-ins for the kernels (CPU oracle, eager assembly, eager network): exactly one exchange per iteration,
    the exchanged gradients are the mean of the ranks' own, and the replicas stay bit-identical through optimizer steps."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_frame_step, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, f0, c0, e0, b0, mine0, red0, w0, gx0, t0), (_, f1, c1, e1, b1, mine1, red1, w1, gx1, t1) = res
    assert f0 == [0, 2, 4] and f1 == [1, 3, 5]                    # frames sharded over the ranks
    assert t0 == 0.0 and abs(t1 - 1 / 7) < 1e-12                  # each rank queries the network at its own frame's time
    assert c0 == c1 == e0 == e1 == 3                              # one collective per iteration, no more
    assert b0 == b1 == (522055 - 5140) * 4
    import numpy as np
    assert not np.array_equal(mine0, mine1)                       # different frames -> different local gradients
    np.testing.assert_allclose(red0, (mine0 + mine1) / 2, rtol=1e-6, atol=1e-12)
    assert np.array_equal(red0, red1)
    assert np.array_equal(w0, w1)                                 # replicas end bit-identical
    assert gx0 > 0 and gx1 > 0 and gx0 != gx1                     # the Gaussians' own gradients stay local (per frame)
