T=${1:-r03_v18}
python3 bench.py > gpurun_out/${T}_bench_full.json 2> gpurun_out/${T}_bench_full.err && echo full ok &&
python3 bench.py --workload C1 --no-cpu-baseline > gpurun_out/${T}_bench_C1.json 2>/dev/null && echo c1 ok &&
python3 bench.py --workload C2 --no-cpu-baseline --extras varying_views > gpurun_out/${T}_bench_C2.json 2>/dev/null && echo c2 ok &&
python3 bench.py --workload C5 --no-cpu-baseline --extras varying_views > gpurun_out/${T}_bench_C5.json 2>/dev/null && echo c5 ok &&
python3 bench.py --workload C3 > gpurun_out/${T}_bench_C3.json 2>/dev/null && echo c3 ok &&
python3 bench.py --workload C3 --pair > gpurun_out/${T}_bench_C3_pair.json 2>/dev/null && echo c3p ok
python3 bench.py --workload C4 --no-cpu-baseline > gpurun_out/${T}_bench_C4.json 2>/dev/null && echo c4 ok
python3 bench.py --workload fog --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_fog.json 2>/dev/null && echo fog ok
python3 profiles/stage_bench.py c3 200 > gpurun_out/${T}_stage_c3.json 2>/dev/null
GFT_FWD_SEG=0 python3 profiles/stage_bench.py c3 200 > gpurun_out/${T}_stage_c3_serial.json 2>/dev/null
python3 - <<PY
import json
for w in ("full","C1","C2","C5","C3","C3_pair","C4","fog"):
    try:
        d=json.loads(open("gpurun_out/${T}_bench_%s.json"%w).read().strip().splitlines()[-1])
    except Exception as e:
        print(w, "ERR", e); continue
    print(w, round(d["value"],1), d["unit"], round(d["ms_per_step"],4), "roofline", d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("counters_belong_to_this_source"))
    ex=d.get("extras",{})
    for k,v in ex.items():
        keys=[x for x in ("it_per_s","hip_ms","hip_it_per_s","hip_pair_ms","ms_per_step") if x in v]
        print("   ",k,{x:(round(v[x],3) if isinstance(v[x],float) else v[x]) for x in keys})
    if "cpu_baseline" in d: print("    cpu", d["cpu_baseline"])
PY
python3 bench.py --workload C3 --graph --no-cpu-baseline > gpurun_out/${T}_bench_C3_graph.json 2>/dev/null && echo c3graph ok
python3 bench.py --workload C3 --torch-loss --no-cpu-baseline > gpurun_out/${T}_bench_C3_torch_loss.json 2>/dev/null && echo c3torch ok
