#!/bin/bash
# SQ wave-state / LDS / memory counters of the Winograd kernel on one layer shape:  tools/pmc_wino.sh <wino_bench layer> [batch] [persist knob]
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/pmcw"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES TA_TA_BUSY TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf "$OUT/p$i"
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/p$i" -o p -- python3 "$REPO/tools/wino_bench.py" --layers "$1" --batch "${2:-1024}" --check 1 --iters 3 --persist "${3:-8}" > "$OUT/p$i.log" 2>&1
  python3 - "$OUT/p$i" winograd_ <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if sys.argv[2] in r["Kernel_Name"]] if f else []
agg = collections.OrderedDict()
for r in rows:
    agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(" ".join(f"{c}={sum(v[-3:])/len(v[-3:]):.4g}" for c, v in agg.items()))
PY
done
