"""One training iteration of the reference's loop shape with every piece from this package (bench.train_iteration_extra,
`only_fused`), for `rocprofv3 --kernel-trace --stats -- python3 profiles/train_workload.py [steps]`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from gftorf_amd import _lib  # noqa: E402

_lib.load()
dev = torch.device("cuda:0")
scene = bench.build_scene("metric", 0, 1)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
print(json.dumps(bench.train_iteration_extra(dev, scene, steps=steps, warmup=3, only_fused=True)))
