"""cProfile of the operator's host path: N forward + backward calls of the tiny scene (20 k Gaussians, 256 x 256: the
kernels take less than the host), top functions by own and by cumulative time.  `python profiles/host_profile.py [N]`."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench
from gftorf_amd import GaussianRasterizationSettings, GaussianRasterizer

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = torch.device("cuda:0")
scene = bench.build_scene("tiny", 0, 1)
cfg, g, cam = scene["cfg"], scene["gaussians"], scene["cam"]
P, W, H = cfg["P"], cfg["W"], cfg["H"]
t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
r = GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], bg=t(scene["bg"]), scale_modifier=1.0,
    viewmatrix=t(cam["viewmatrix"]), projmatrix=t(cam["projmatrix"]), sh_degree=cfg["D"], campos=t(cam["campos"]), prefiltered=False, debug=False,
    near_n=cam["znear"], far_n=cam["zfar"], depth_range=scene["depth_range"], use_view_dependent_phase=scene["use_view_dependent_phase"]))
leaf = {k: t(v).requires_grad_(True) for k, v in g.items() if v is not None}
m2 = torch.zeros((P, 3), device=dev, requires_grad=True)
gr = {k: t(v) for k, v in scene["grads"].items()}
ups = [gr["color"], gr["phasor"], gr["depth"], gr["acc"], gr["depth_distortion"]]


def step():
    for x in leaf.values():
        x.grad = None
    m2.grad = None
    o = r(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"], scales=leaf["scales"],
          rotations=leaf["rotations"], phase_offset=scene["phase_offset"], dc_offset=scene["dc_offset"])
    torch.autograd.backward([o[0], o[1], o[2], o[4], o[6]], ups)


for _ in range(200):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host us per forward + backward (unprofiled): %.1f" % ((t1 - t0) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(35)
    txt = s.getvalue()
    print("\n".join(l for l in txt.splitlines() if l.strip())[:6000])
    print("(per call: divide the times by %d)" % N)
