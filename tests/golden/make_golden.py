"""Generate tests/golden/*.npz from the reference's importable Python helpers.

Runs ONLY in the authoring container (needs /root/reference).  The reference
files themselves never travel; only the input/output vectors written here do.

  sh_color.npz   : utils/sh_utils.py eval_sh (lines 57-112) on random
                   coefficients/directions, degrees 0..3
  camera.npz     : utils/graphics_utils.py getProjectionMatrix (55-75),
                   getProjectionMatrixShift (77-109), getWorld2View2 (38-49)
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from utils.sh_utils import eval_sh, RGB2SH, PA2SH  # noqa: E402
from utils.graphics_utils import (getProjectionMatrix, getProjectionMatrixShift,  # noqa: E402
                                  getWorld2View2)

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20251003)
    n = 64
    sh = rng.normal(0, 0.5, (n, 16, 3)).astype(np.float32)       # kernel layout [coeff][channel]
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out = {}
    for deg in range(4):
        # helper wants [..., C, coeff]
        res = eval_sh(deg, torch.tensor(sh).permute(0, 2, 1), torch.tensor(dirs))
        out[f"deg{deg}"] = res.numpy().astype(np.float32)
    x = rng.random((8, 3)).astype(np.float32)
    np.savez(os.path.join(HERE, "sh_color.npz"), sh=sh, dirs=dirs,
             rgb2sh_in=x, rgb2sh_out=RGB2SH(torch.tensor(x)).numpy(),
             pa2sh_out=PA2SH(torch.tensor(x)).numpy(), **out)

    cams = {}
    for name, (W, H, fovx_deg, zn, zf) in {
            "bench640": (640, 480, 60.0, 0.45, 6.05),
            "bench1080": (1920, 1080, 60.0, 0.45, 6.05),
            "c1_256": (256, 256, 60.0, 0.45, 6.05)}.items():
        tanx = math.tan(math.radians(fovx_deg) * 0.5)
        tany = tanx * H / W
        fovx, fovy = 2 * math.atan(tanx), 2 * math.atan(tany)
        cams[name + "_proj"] = getProjectionMatrix(zn, zf, fovx, fovy).numpy()
        cams[name + "_args"] = np.array([W, H, fovx, fovy, zn, zf], np.float64)
    W, H, fx, fy, cx, cy = 320, 240, 260.0, 262.0, 155.5, 118.25
    fovx, fovy = 2 * math.atan(W / (2 * fx)), 2 * math.atan(H / (2 * fy))
    cams["shift_proj"] = getProjectionMatrixShift(0.45, 6.05, fx, fy, cx, cy, W, H, fovx, fovy).numpy()
    cams["shift_args"] = np.array([0.45, 6.05, fx, fy, cx, cy, W, H, fovx, fovy], np.float64)
    # world-to-view for a non-trivial pose
    a = 0.3
    R = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
    t = np.array([0.1, -0.2, 0.3])
    cams["w2v_R"], cams["w2v_t"] = R, t
    cams["w2v"] = getWorld2View2(R, t)
    np.savez(os.path.join(HERE, "camera.npz"), **cams)
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
