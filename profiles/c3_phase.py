"""Device and wall time of one C3 iteration by phase (`python3 profiles/c3_phase.py warm|net [iterations]`): the loop of
bench_loop.py with the deformation network off (warm: the first 2000 iterations of the config) or on from the first
iteration (net).  Prints wall ms per iteration; run under `rocprofv3 --kernel-trace --stats` for the kernel side."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch              # noqa: E402
import bench_loop         # noqa: E402

phase = sys.argv[1] if len(sys.argv) > 1 else "warm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
graph = "--graph" in sys.argv
cfg = dict(bench_loop.C3, warm_up=0 if phase == "net" else 10 ** 9)
dev = torch.device("cuda:0")
iteration, info = bench_loop.build_loop(dev, cfg, graph=graph, fused_loss="--torch-loss" not in sys.argv)
for it in range(1, 41):
    iteration(it)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(41, 41 + n):
    iteration(it)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"phase": phase, "graph": graph, "iterations": n, "wall_ms_per_iteration": dt / n * 1e3}))
