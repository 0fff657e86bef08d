#!/usr/bin/env python3
"""Workload for the --pmc passes of the pair evidence (profiles/collect_pair.sh): N iterations of the colour + ToF camera
forward + backward at the metric size, either as two GaussianRasterizer calls whose gradients autograd adds ("two") or as
one GaussianRasterizerPair call ("pair").  usage: pair_workload.py two|pair [iterations]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402
import bench  # noqa: E402
from gftorf_amd import _lib  # noqa: E402

_lib.load()
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda:0")
scene = bench.build_scene("metric", 0, 1)
r = bench.pair_extra(dev, scene, steps=n, warmup=5, which=(which,))
torch.cuda.synchronize()
print(which, r)
