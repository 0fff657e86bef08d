"""Host time of one forward + backward through the public API, split into the two C calls (kernel launches + the wait for
the forward's stage-1 totals) and everything else (Python + torch autograd): `python profiles/host_split.py`."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from gftorf_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib, api
lib = _lib.load()
acc = {"gft_forward": 0.0, "gft_backward": 0.0, "n": 0}
for name in ("gft_forward", "gft_backward"):
    orig = getattr(lib, name)
    def wrap(*a, _o=orig, _n=name):
        t = time.perf_counter()
        r = _o(*a)
        acc[_n] += time.perf_counter() - t
        return r
    setattr(lib, name, wrap)
dev = torch.device("cuda:0")
scene = bench.build_scene("tiny", 0, 1)
cfg, g = scene["cfg"], scene["gaussians"]
P, W, H = cfg["P"], cfg["W"], cfg["H"]
t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
cam = scene["cam"]
r = GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(scene["bg"]), scale_modifier=1.0,
    viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=cfg["D"], campos=t(cam["campos"]), prefiltered=False, debug=False,
    near_n=cam["znear"], far_n=cam["zfar"], depth_range=scene["depth_range"], use_view_dependent_phase=scene["use_view_dependent_phase"]))
leaf = {k: t(v).requires_grad_(True) for k, v in g.items() if v is not None}
m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
gr = {k: t(v) for k, v in scene["grads"].items()}
ups = [gr["color"], gr["phasor"], gr["depth"], gr["acc"], gr["depth_distortion"]]
tf = tb = 0.0
def step(timed=False):
    global tf, tb
    for x in leaf.values(): x.grad = None
    m2.grad = None
    t0 = time.perf_counter()
    o = r(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"], scales=leaf["scales"],
          rotations=leaf["rotations"], phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"])
    t1 = time.perf_counter()
    torch.autograd.backward([o[0], o[1], o[2], o[4], o[6]], ups)
    t2 = time.perf_counter()
    if timed:
        tf += t1 - t0; tb += t2 - t1
for _ in range(100): step()
torch.cuda.synchronize()
acc.update(gft_forward=0.0, gft_backward=0.0)
N = 1000
t0 = time.perf_counter()
for _ in range(N): step(True)
t1 = time.perf_counter()
torch.cuda.synchronize()
us = lambda x: round(x / N * 1e6, 1)
print({"host_us_per_step": us(t1 - t0), "forward_call_us": us(tf), "of_which_gft_forward_C_us": us(acc["gft_forward"]),
       "backward_call_us": us(tb), "of_which_gft_backward_C_us": us(acc["gft_backward"]), "scene": "tiny (20 k Gaussians, 256 x 256): the kernels take less than the host"})
