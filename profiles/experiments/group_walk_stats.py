"""The statistic in front of round 6's sub-group walk of the blend kernels: if the 64 lanes of a quadrant
wave are split into G groups that each walk their OWN hits of a staged batch, how many passes does a batch take against
today's one pass per entry that any pixel blends?  CPU only: the oracle's forward gives the lists and the per-pixel
contributor counts, group_walk_stats.c counts.  `python profiles/experiments/group_walk_stats.py [metric fog c3frame C5 C2]`"""
import ctypes as C
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                      # noqa: E402
from gftorf_amd import synth            # noqa: E402
from oracle import oracle               # noqa: E402
import helpers as Hh                    # noqa: E402

so = os.path.join("/tmp", "group_walk_stats.so")
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "group_walk_stats.c"), "-lm"])
L = C.CDLL(so)
GROUPS = [(1, "8x8 (today)"), (2, "8x4"), (4, "4x4"), (8, "4x2"), (16, "2x2"), (64, "lane")]
BATCH = [64, 128, 256]


def scene(name):
    if name == "c3frame":
        # the reference's own size (configs/torf.json: 100 k Gaussians, 320x240) at the opacity its scenes start from
        cam = synth.make_camera(320, 240)
        g = synth.make_gaussians(100_000, cam, 1236, sh_coeffs=16, scale_lo=0.004, scale_hi=0.04, opacity_range=(0.1, 0.1))
        cfg = dict(P=100_000, W=320, H=240, D=3, sh_coeffs=16, tof=True)
        return dict(cfg=cfg, cam=cam, gaussians=g, bg=synth.make_background(320, 240, 1236), grads=synth.make_pixel_grads(320, 240, 1236),
                    depth_range=10.0, phase_offset=0.1, dc_offset=0.05, use_view_dependent_phase=True)
    return synth.make_scene(name)


res = {}
for name in ([a for a in sys.argv[1:] if not a.startswith("--")] or ["metric", "fog", "c3frame"]):
    sc = scene(name)
    f, _ = Hh.run_oracle(oracle, sc, backward=False)
    g = f.geom
    p = lambda a: C.c_void_p(a.ctypes.data)
    arrs = [np.ascontiguousarray(x) for x in (f.ranges, np.asarray(f.point_list, np.uint32), g["means2D"], g["conic_opacity"], f.img["n_contrib"])]
    out = np.zeros(5 + 3 * len(GROUPS) * 3)
    L.gfto_group_walk_stats(C.c_int(f.W), C.c_int(f.H), *[p(a) for a in arrs], p(out))
    q, walked, hits, hits_cull, pairs = out[:5]
    r = {"quadrants": int(q), "walked_per_quadrant": walked / q, "wave_hits_live_per_quadrant": hits / q,
         "wave_hits_cull_per_quadrant": hits_cull / q, "lanes_per_live_hit": pairs / hits, "batches": {}}
    print("%s: %d quadrants, %.0f entries walked, %.1f reach the quadrant by the cull, %.1f blended by some pixel, %.1f of 64 lanes per hit"
          % (name, q, walked / q, hits_cull / q, hits / q, pairs / hits))
    for b, B in enumerate(BATCH):
        row = {}
        for gi, (G, label) in enumerate(GROUPS):
            pl, pc, sc_ = out[5 + (b * len(GROUPS) + gi) * 3: 5 + (b * len(GROUPS) + gi) * 3 + 3]
            row[label] = {"passes_live_per_wave_hit": pl / hits, "passes_cull_per_wave_hit": pc / hits,
                          "group_records_per_wave_hit": sc_ / hits}
        r["batches"][B] = row
        print("  batch %3d: passes per today's pass (by the cull | floor by live pixels): " % B +
              ", ".join("%s %.3f | %.3f" % (lab, row[lab]["passes_cull_per_wave_hit"], row[lab]["passes_live_per_wave_hit"]) for _, lab in GROUPS))
        if B == 64:
            print("             mean over the groups (what perfectly balanced groups would take): " +
                  ", ".join("%s %.3f" % (lab, row[lab]["group_records_per_wave_hit"] / G) for G, lab in GROUPS))
    res[name] = r
if "--json" in sys.argv:
    print(json.dumps(res))
