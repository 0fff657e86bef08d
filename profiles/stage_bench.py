"""Per-stage times (HIP events inside the library) of the rasterizer's forward + backward on one frame:
`python profiles/stage_bench.py <frame> [steps]`, frame = c3 (100 k Gaussians at the reference's initial opacity 0.1,
320x240: the shape of configs/torf.json), metric, fog, c2, c5.  For A/B runs of kernel variants (environment switches)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np     # noqa: E402
import torch           # noqa: E402
from gftorf_amd import _lib, api, synth   # noqa: E402
if os.environ.get("GFT_ABL_LIB"):          # an experiment build (python -m gftorf_amd.build --tag ...), as profiles/bench_with_lib.py
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["GFT_ABL_LIB"])
import helpers as Hh   # noqa: E402


def frame(name):
    if name == "c3":
        W, H, P = 320, 240, 100_000
        cam = synth.make_camera(W, H)
        g = synth.make_gaussians(P, cam, 1236, sh_coeffs=16, scale_lo=0.004, scale_hi=0.04, opacity_range=(0.1, 0.1))
        return dict(cfg=dict(P=P, W=W, H=H, D=3, sh_coeffs=16, tof=True), cam=cam, gaussians=g, bg=synth.make_background(W, H, 5),
                    grads=synth.make_pixel_grads(W, H, 5), depth_range=10.0, phase_offset=0.1, dc_offset=0.02,
                    use_view_dependent_phase=True)
    return synth.make_scene({"metric": "metric", "fog": "fog", "c2": "C2", "c5": "C5"}[name])


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "c3"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dev = torch.device("cuda:0")
    sc = frame(name)
    g, cfg = sc["gaussians"], sc["cfg"]
    t = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    from gftorf_amd import GaussianRasterizer
    rast = GaussianRasterizer(Hh.gpu_settings(sc, dev))
    leaf = {k: t(v).requires_grad_(True) for k, v in g.items() if v is not None}
    m2 = torch.zeros((cfg["P"], 3), device=dev, requires_grad=True)
    ups = [t(sc["grads"][k]) for k in ("color", "phasor", "depth", "acc", "depth_distortion")]

    def step():
        for x in leaf.values():
            x.grad = None
        m2.grad = None
        o = rast(means3D=leaf["means3D"], means2D=m2, opacities=leaf["opacities"], shs=leaf["shs"], shs_p=leaf["shs_p"],
                 scales=leaf["scales"], rotations=leaf["rotations"], phase_offset=sc["phase_offset"], dc_offset=sc["dc_offset"])
        torch.autograd.backward([o[0], o[1], o[2], o[4], o[6]], ups)
        return o
    for _ in range(20):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    _lib.profile_reset()
    _lib.profile_enable(True)
    for _ in range(steps):
        o = step()
    torch.cuda.synchronize()
    prof = _lib.profile_read()
    _lib.profile_enable(False)
    calls = max(prof["forward_calls"], 1)
    st = {k[:-3]: round(prof[k] / calls * 1e3, 1) for k in prof if k.endswith("_ms")}
    print(json.dumps({"frame": name, "ms_per_step": round(dt * 1e3, 4), "stage_us": st, "sum_us": round(sum(st.values()), 1),
                      "R": api.last_call_stats["num_rendered"], "blended": int((o[8] > 0).sum()),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("GFT_")}}), flush=True)


if __name__ == "__main__":
    main()
