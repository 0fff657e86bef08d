import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gftorf_amd import _lib, api, synth
import helpers as Hh
sc = synth.make_scene(sys.argv[1] if len(sys.argv) > 1 else "metric")
dev = torch.device("cuda:0")
g, cfg = sc["gaussians"], sc["cfg"]
t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
from gftorf_amd import GaussianRasterizer
rast = GaussianRasterizer(Hh.gpu_settings(sc, dev))
leaf = {k: t(v).requires_grad_(True) for k, v in g.items() if v is not None}
m2 = torch.zeros((cfg["P"], 3), device=dev, requires_grad=True)
ups = [t(sc["grads"][k]) for k in ("color", "phasor", "depth", "acc", "depth_distortion")]
log = []
for it in range(40):
    for x in leaf.values(): x.grad = None
    m2.grad = None
    o = rast(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"], scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=sc["phase_offset"], dc_offset=sc["dc_offset"])
    torch.autograd.backward([o[0], o[1], o[2], o[4], o[6]], ups)
    pool = next(iter(api._grad_pool.values()))
    log.append((api.last_call_stats["grads_reused"], api.last_call_stats["grads_rows_only"], [int(e["report_np"][0]) for e in pool], [e["dense_left"] for e in pool], len(pool)))
torch.cuda.synchronize()
for l in log: print(l)
