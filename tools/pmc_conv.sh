#!/bin/bash
# SQ wave-state counters of single conv layers (tools/conv_bench.py) — where do the waves of a short-K 1x1 GEMM spend their time?
#   tools/pmc_conv.sh <layers> [batch]   -> gpurun_out/r03/pmc_conv_<n>.csv (per-dispatch counters)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/r03"; mkdir -p "$OUT"; export TMPDIR=/tmp
L="$1"; B="${2:-1024}"
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES TA_TA_BUSY GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf "$OUT/pmcc$i"
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmcc$i" -o p -- python3 "$REPO/tools/conv_bench.py" --batch "$B" --iters 2 --layers "$L" > "$OUT/pmcc$i.log" 2>&1
  python3 - "$OUT/pmcc$i" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counter file", sys.argv[1]); sys.exit()
rows = list(csv.DictReader(open(f[0])))
# last 2 dispatches of each (kernel, grid) = the timed iterations
agg = collections.OrderedDict()
for r in rows:
    k = r["Kernel_Name"]
    if "vatl::" not in k or "pack" in k: continue
    key = (k.split("(")[0][-60:], r["Grid_Size"])
    agg.setdefault(key, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for key, cs in agg.items():
    print(key[0], "grid", key[1], " ".join(f"{c}={sum(v[-2:])/len(v[-2:]):.4g}" for c, v in cs.items()))
PY
done
find "$OUT" -name "*.csv" -size +20M -delete
