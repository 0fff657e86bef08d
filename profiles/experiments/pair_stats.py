"""Statistics behind DESIGN.md's bound for k_render_bwd: how many (quadrant, splat) hits of the backward's walk could
share a pass (hits whose pixel masks are disjoint commute)?  CPU only: the oracle's forward gives the lists and the
per-pixel contributor counts.  `python profiles/experiments/pair_stats.py [workload]`"""
import ctypes as C
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

so = os.path.join("/tmp", "pair_stats.so")
subprocess.check_call(["gcc", "-O2", "-fopenmp", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "pair_stats.c"), "-lm"])
L = C.CDLL(so)
name = sys.argv[1] if len(sys.argv) > 1 else "metric"
sc = synth.make_scene(name)
f, _ = Hh.run_oracle(oracle, sc, backward=False)
g = f.geom
p = lambda a: C.c_void_p(a.ctypes.data)
arrs = [np.ascontiguousarray(x) for x in (f.ranges, np.asarray(f.point_list, np.uint32), g["means2D"], g["conic_opacity"], f.img["n_contrib"])]
for w in (4, 8):
    out = np.zeros(8)
    L.gfto_pair_stats(C.c_int(f.W), C.c_int(f.H), *[p(a) for a in arrs], C.c_int(w), p(out))
    h = out[0]
    q = 4 * ((f.W + 15) // 16) * ((f.H + 15) // 16)
    print("%s, window %d: %.1f hits and %.1f walked entries per quadrant, %.1f of 64 lanes per hit; passes per hit: two consecutive "
          "%.3f, four consecutive %.3f, pairs within the window %.3f, four within the window %.3f"
          % (name, w, h / q, out[6] / q, out[1] / h, out[2] / h, out[3] / h, out[4] / h, out[5] / h))
