#!/bin/bash
# SQ counters of the F(4x4,3x3) Winograd kernel (and, for comparison, of the F(2x2) kernel of the same launch):  tools/pmc_f4.sh [layer] [crops]
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/f4"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_BUSY_CU_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf "$OUT/pmc$i"
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OUT/pmc$i" -o p -- python3 "$REPO/tools/f4_bench.py" --layers "${1:-l3.c2}" --batch "${2:-1024}" --iters 3 > "$OUT/pmc$i.log" 2>&1
  for k in winograd_f4_kernel "winograd_kernel<2"; do
  python3 - "$OUT/pmc$i" "$k" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if sys.argv[2] in r["Kernel_Name"]] if f else []
agg = collections.OrderedDict()
for r in rows:
    agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(sys.argv[2], " ".join(f"{c}={sum(v[-3:])/len(v[-3:]):.4g}" for c, v in agg.items()))
PY
  done
done
