"""Where does the headline's timed region go when its mean step is above the median step?
`python3 profiles/experiments/host_device_gap.py [legs] [steps]`: the metric step of bench.py, `legs` timed regions of
`steps` steps as the driver runs them (sync, K steps, sync), for each: wall per step, host enqueue time per step (perf_counter
around each step() call, no sync), device time per step (one event per step), and the three largest host / device steps."""
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch              # noqa: E402
import bench              # noqa: E402

legs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
scene = bench.build_scene("metric", 0, 1)
step, state, _ = bench.gpu_step_fn(scene, dev)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.5:
    step()
torch.cuda.synchronize()
out = []
for leg in range(legs):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    host = []
    gc0 = gc.get_count()
    w0 = time.perf_counter()
    marks[0].record()
    for i in range(steps):
        h0 = time.perf_counter()
        step()
        marks[i + 1].record()
        host.append((time.perf_counter() - h0) * 1e3)
    h_end = time.perf_counter()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - w0) * 1e3
    devs = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    out.append(dict(leg=leg, wall_ms_per_step=wall / steps, host_ms_per_step=sum(host) / steps, host_done_before_end_ms=(wall - (h_end - w0) * 1e3),
                    dev_ms_per_step=sum(devs) / steps, dev_median=sorted(devs)[steps // 2], host_top3=sorted(host)[-3:], dev_top3=sorted(devs)[-3:],
                    first_dev=devs[0], gc_count=gc0))
for o in out:
    print(json.dumps({k: (round(v, 4) if isinstance(v, float) else ([round(x, 3) for x in v] if isinstance(v, list) else v)) for k, v in o.items()}))
