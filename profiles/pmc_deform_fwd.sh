#!/bin/bash
# Counters of the deformation network's forward kernel (saving and inference shapes): pmc_deform_fwd.sh <outdir>
# Separate passes: a pass whose counter names this rocprofv3 does not know fails alone.
set -u
OUT=${1:-gpurun_out/pmc_dfwd}
export TMPDIR=/tmp
mkdir -p "$OUT"
cat > "$OUT/wl.py" <<'PY'
import sys, torch
sys.path.insert(0, ".")
from gftorf_amd import reference_network
from oracle import deform_ref
dev = torch.device("cuda:0")
params = deform_ref.random_params(3)
net = reference_network(); net.load_state_dict({k: torch.tensor(v) for k, v in params.items()}); net = net.to(dev)
n = 300000
x = torch.rand((n, 3), device=dev); t = torch.full((1, 1), 0.4, device=dev).expand(n, -1)
for _ in range(3):
    a = net(x, t)                 # saving shape
    with torch.no_grad():
        b = net(x, t)             # inference shape
torch.cuda.synchronize()
PY
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F16" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -- python3 "$OUT/wl.py" > "$OUT/p$i.log" 2>&1
    echo "pass $i rc=$? ($set)"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_deform_fwd_h" not in k:          # (not the bf16 walk's returning launches behind it)
            continue
        k = "saving (k_deform_fwd_h<true, 4>)" if "Lb1" in k or "<true" in k else "inference (k_deform_fwd_h<false, 8>)"
        res[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_VALU", "SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM", "TCP_TCC_READ_REQ_sum", "TCC_HIT_sum"):
            res[k]["dur_us:" + r["Counter_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
summ = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in res.items()}
json.dump(summ, open(out + "/deform_fwd_counters.json", "w"), indent=1)
for k, d in summ.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-34s %.4g" % (c, v))
PY
