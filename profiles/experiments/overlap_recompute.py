"""Ceiling of the last review's item 6 ("FrameStep: the on-demand recompute on a second stream beside the rasterizer's
backward"), measured before building it: the metric frame's rasterizer backward on the main stream and the deformation
network's saving forward over 20 k rows (what the recompute is in the C4 step) on a side stream, started together, against
the two one after the other on one stream.  `python profiles/experiments/overlap_recompute.py [rows]` -> one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np     # noqa: E402
import torch           # noqa: E402
from gftorf_amd import synth, reference_network, GaussianRasterizer   # noqa: E402
import helpers as Hh   # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000
dev = torch.device("cuda:0")
sc = synth.make_scene("metric")
g, cfg = sc["gaussians"], sc["cfg"]
t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
rast = GaussianRasterizer(Hh.gpu_settings(sc, dev))
leaf = {k: t(v).requires_grad_(True) for k, v in g.items() if v is not None}
m2 = torch.zeros((cfg["P"], 3), device=dev, requires_grad=True)
ups = [t(sc["grads"][k]) for k in ("color", "phasor", "depth", "acc", "depth_distortion")]
net = reference_network().to(dev)
x = torch.rand((rows, 3), device=dev)
tt = torch.full((1, 1), 0.4, device=dev).expand(rows, -1)
side = torch.cuda.Stream(device=dev)


def forward():
    for v in leaf.values():
        v.grad = None
    m2.grad = None
    return rast(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
                scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=sc["phase_offset"], dc_offset=sc["dc_offset"])


def timed(mode, n=60):
    tot = 0.0
    for i in range(n + 10):
        o = forward()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        if mode == "serial":
            torch.autograd.backward([o[0], o[1], o[2], o[4], o[6]], ups)
            net(x, tt)                                   # (a saving forward: the module is in training mode, grads enabled)
        elif mode == "raster":
            torch.autograd.backward([o[0], o[1], o[2], o[4], o[6]], ups)
        elif mode == "network":
            net(x, tt)
        else:
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                net(x, tt)
            torch.autograd.backward([o[0], o[1], o[2], o[4], o[6]], ups)
            torch.cuda.current_stream(dev).wait_stream(side)
        b.record()
        torch.cuda.synchronize()
        if i >= 10:
            tot += a.elapsed_time(b)
    return tot / n


res = {m: round(timed(m), 4) for m in ("raster", "network", "serial", "overlapped", "serial", "overlapped")}
print(json.dumps({"rows": rows, "ms": res, "note": "raster = the metric frame's backward alone, network = the saving forward over `rows` alone"}))
