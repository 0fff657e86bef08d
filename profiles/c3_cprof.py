"""Host-side profile of the C3 loop (bench.py --workload C3): cProfile over 600 iterations (100 warm-up) -- which
Python functions the iteration's wall time goes to.  `python3 profiles/c3_cprof.py [pattern]`."""
import cProfile
import io
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pattern = sys.argv[1] if len(sys.argv) > 1 else ""
sys.argv = ["bench.py", "--workload", "C3", "--steps", "600", "--warmup", "100", "--no-cpu-baseline", "--no-extras"]
import bench  # noqa: E402

pr = cProfile.Profile()
pr.enable()
try:
    bench.main()
except SystemExit:
    pass
pr.disable()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(pattern, 45) if pattern else ps.print_stats(45)
print(s.getvalue()[:9000])
