"""Per-stage times of the CPU oracle (the `cpu_baseline` of bench.py) for a few thread counts, with the host's CPU
budget beside them: `python profiles/oracle_stage_times.py [workload] [threads ...]`."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from gftorf_amd import synth          # noqa: E402
from oracle import oracle             # noqa: E402
import helpers as Hh                  # noqa: E402

oracle.build()
name = sys.argv[1] if len(sys.argv) > 1 else "metric"
threads = [int(a) for a in sys.argv[2:]] or [oracle.num_threads()]
info = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try:
        info[f] = open(f).read().strip()
    except OSError:
        pass
print(json.dumps(info), flush=True)
sc = synth.make_scene(name)
L = oracle.lib()
times = {}


class Timed:
    def __init__(self, n, f):
        self.n, self.f = n, f

    def __call__(self, *a):
        t = time.perf_counter()
        r = self.f(*a)
        times[self.n] = times.get(self.n, 0.0) + time.perf_counter() - t
        return r


for n in ["gfto_preprocess_fwd", "gfto_scan", "gfto_duplicate_with_keys", "gfto_sort_pairs", "gfto_tile_ranges", "gfto_render_fwd",
          "gfto_render_bwd", "gfto_unpack_acc", "gfto_preprocess_bwd", "gfto_zero"]:
    setattr(L, n, Timed(n, getattr(L, n)))
oracle.reuse_buffers(True)
for nt in threads:
    oracle.set_num_threads(nt)
    for it in range(3):
        times.clear()
        t = time.perf_counter()
        Hh.run_oracle(oracle, sc)
        tot = time.perf_counter() - t
    print(json.dumps({"threads": nt, "total_s": round(tot, 4), "stages_s": {k[5:]: round(v, 4) for k, v in times.items()},
                      "python_s": round(tot - sum(times.values()), 4)}), flush=True)
