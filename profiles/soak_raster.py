"""Randomised soak of the rasterizer (public API -> C ABI -> kernels) against the C oracle with the parity tests' own
checks; the generator and the mode product it walks live in tests/soak_cases.py (a seeded 200-frame slice of it runs in the
GPU suite: tests/test_gpu_soak.py).  `python profiles/soak_raster.py [seconds] [seed]` -> one JSON line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import soak_cases
from oracle import oracle as O

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 77
O.build(); O.lib()
print(json.dumps(soak_cases.run(torch.device("cuda:0"), O, seed=seed, seconds=budget)))
