"""Statistic behind VERDICT r4 #3 (k_render_fwd: split only what is still open): how long are the forward's serial
chains?  Per 8x8 quadrant of a frame, the list position of its deepest contributor (the forward's wave walks a little
further: until its last pixel ends) from the CPU oracle's n_contrib, and the share of the frame's quadrants / of the
chain work that lies beyond K entries.  CPU only.  `python profiles/experiments/fwd_walk_stats.py [metric|fog|C5] [view]`"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                      # noqa: E402
from gftorf_amd import synth            # noqa: E402
from oracle import oracle               # noqa: E402
import helpers as Hh                    # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "metric"
view = float(sys.argv[2]) if len(sys.argv) > 2 else None
w2c = None
if view is not None:                    # a camera of bench.py's 30-view arc: a in [-0.15, 0.15]
    a = view
    w2c = synth.look_at_w2c(yaw=a, pitch=0.04 * np.sin(3 * a), t=(-3.2 * np.sin(a), 0.0, 3.2 * (1 - np.cos(a))))
sc = synth.make_scene(name)
if w2c is not None:                     # the same Gaussians from another camera (make_scene places them in front of ITS camera)
    sc["cam"] = synth.make_camera(sc["cfg"]["W"], sc["cfg"]["H"], w2c=w2c)
oracle.build()
f, _ = Hh.run_oracle(oracle, sc, backward=False)
W, H = f.W, f.H
gx, gy = (W + 15) // 16, (H + 15) // 16
nc = np.zeros((gy * 16, gx * 16), np.int64)
nc[:H, :W] = np.asarray(f.img["n_contrib"]).reshape(H, W)
Tf = np.ones((gy * 16, gx * 16), np.float64)
Tf[:H, :W] = np.asarray(f.img["final_T"]).reshape(H, W)
q = nc.reshape(gy * 2, 8, gx * 2, 8).max(axis=(1, 3))                       # deepest contributor per quadrant
lens = (f.ranges[:, 1] - f.ranges[:, 0]).reshape(gy, gx)
qlen = np.repeat(np.repeat(lens, 2, 0), 2, 1)                               # list length of the quadrant's tile
open_ = (Tf.reshape(gy * 2, 8, gx * 2, 8) > 0.02).any(axis=(1, 3))         # some pixel far from the 1e-4 stop: the wave walks the whole list
walk = np.where(open_, qlen, np.minimum(qlen, q + 8))
n = walk.size
tot = walk.sum()
print("%s%s: %d quadrants, list length mean %.0f max %d; walk per quadrant: mean %.0f, median %.0f, p90 %.0f, p99 %.0f, max %d; quadrants that walk their whole list: %.1f %%"
      % (name, "" if view is None else " view %.2f" % view, n, qlen.mean(), qlen.max(), walk.mean(), np.median(walk), np.percentile(walk, 90), np.percentile(walk, 99), walk.max(), 100.0 * open_.mean()))
for K in (128, 256, 384, 512, 768, 1024, 2048):
    over = walk > K
    print("  K = %4d: %5.1f %% of the quadrants walk further; the entries beyond K are %5.1f %% of all walked entries; longest remainder %d"
          % (K, 100.0 * over.mean(), 100.0 * np.maximum(walk - K, 0).sum() / tot, max(int(walk.max()) - K, 0)))
