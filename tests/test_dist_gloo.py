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
    port = 29500 + os.getpid() % 2000
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


def test_algorithmic_bytes_formula():
    sys.path.insert(0, ROOT)
    import bench
    per, whole = bench.algorithmic_bytes(P=1_000_000, P_vis=910_000, R=3_000_000, N=640 * 480, T=1200)
    # 1516*P_vis + 432*P_cull + 268*R + 224*N ~ 2.3 GB
    assert whole == 1516 * 910_000 + 432 * 90_000 + 268 * 3_000_000 + 224 * 307_200
    assert abs(whole / 1e9 - 2.29) < 0.05
    # the per-kernel split adds up to the whole-path figure
    assert sum(per.values()) == whole
