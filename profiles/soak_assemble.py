"""Randomised soak of the fused input assembly against the eager restatement of gaussian_renderer/__init__.py:81-105:
random sizes, SH widths, mask densities (none / all dynamic too), offsets as tensors / floats / the network's mix, render
regions, and `rotation` given or None (static rows normalised inside).  `python profiles/soak_assemble.py [seconds]`."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_assemble as TA
from oracle import assemble_ref as R
from gftorf_amd import assemble_inputs
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
t0 = time.time(); n = 0; fused = 0
while time.time() - t0 < budget:
    P = int(rng.choice([1, 2, 63, 64, 65, 1000, 4097, 30000]))
    M, M_p = [(16, 16), (4, 4), (1, 1), (9, 16), (16, 4)][int(rng.integers(5))]
    frac = float(rng.choice([0.0, 0.05, 0.3, 1.0]))
    offsets = ["tensor", "float", "net"][int(rng.integers(3))]
    regions = [("static", "dynamic"), ("static",), ("dynamic",)][int(rng.integers(3))]
    c = TA.make_case(P, M=M, M_p=M_p, frac=frac, seed=int(rng.integers(1 << 30)), offsets=offsets)
    no_rot = bool(rng.random() < 0.5)
    def eager(*args, **kw):
        a = list(args)
        if no_rot:
            a[4] = torch.nn.functional.normalize(a[5])
        return R.assemble_eager(*a, **kw)
    def hip(*args, **kw):
        a = list(args)
        if no_rot:
            a[4] = None
        return assemble_inputs(*a, **kw)
    ref, rg = TA.run(eager, dict(c), "cpu", regions)
    got, gg = TA.run(hip, c, dev, regions, validate=True)
    for name, a, b in zip(TA.OUTS, ref, got):
        if name == "rotations":
            np.testing.assert_allclose(b, a, rtol=3e-7, atol=1e-7, err_msg=name)
        else:
            np.testing.assert_array_equal(b, a, err_msg=name)
    for k in TA.ORDER:
        if rg[k] is None or (no_rot and k == "rotation"):
            continue
        if k in ("rotation_raw", "d_rot"):
            np.testing.assert_allclose(gg[k], rg[k], rtol=2e-5, atol=2e-6, err_msg=k)
        else:
            np.testing.assert_array_equal(gg[k], rg[k], err_msg=k)
    n += 1; fused += int(no_rot)
print(json.dumps({"seconds": round(time.time() - t0, 1), "cases": n, "with_rotation_None": fused,
                  "checks": "copies and single adds bit-exact, rotations 3e-7, their gradients 2e-5"}))
