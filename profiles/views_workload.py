"""The varying-view step alone (bench.views_extra: 30 views on an arc in shuffled order, forward + backward), for
`rocprofv3 --kernel-trace --stats -- python3 profiles/views_workload.py [workload] [steps]`."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import bench  # noqa: E402
from gftorf_amd import _lib, api  # noqa: E402

_lib.load()
api.keep_last_buffers = True
dev = torch.device("cuda:0")
scene = bench.build_scene(sys.argv[1] if len(sys.argv) > 1 else "metric", 0, 1)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
print(json.dumps(bench.views_extra(dev, scene, steps=steps, warmup=30)))
