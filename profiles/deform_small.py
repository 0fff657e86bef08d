"""Deformation network forward time against the number of points (latency of one workgroup vs throughput)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gftorf_amd import reference_network
dev = torch.device("cuda:0")
net = reference_network().to(dev)
def timed(fn, k=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(k): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / k
from gftorf_amd import deform as D
D.lazy_save = False
for n in (64, 1024, 8192, 16384, 20000, 32768, 65536, 131072, 300000):
    x = torch.rand((n, 3), device=dev); t = torch.full((1, 1), 0.4, device=dev).expand(n, -1)
    def inf():
        with torch.no_grad(): net(x, t)
    def sav():
        net(x, t)
    print(json.dumps({"points": n, "inference_ms": round(timed(inf), 4), "saving_ms": round(timed(sav), 4)}))
