"""Runs the deformation network's saving forward back to back for about ten seconds (for clock / power sampling)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gftorf_amd import reference_network
dev = torch.device("cuda:0")
net = reference_network().to(dev)
n = 300_000
x = torch.rand((n, 3), device=dev); t = torch.full((1, 1), 0.4, device=dev).expand(n, -1)
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
t0 = time.time()
while time.time() - t0 < 11:
    for _ in range(50):
        if mode == "fwd":
            with torch.no_grad():
                net(x, t)
        else:
            time.sleep(0.01)
    torch.cuda.synchronize()
