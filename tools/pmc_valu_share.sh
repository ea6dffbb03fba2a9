#!/bin/bash
# Matrix-pipe share and vector-instruction share of every kernel of one headline step (they are additive on this hardware: profiles/r04_notes.md):
#   tools/pmc_valu_share.sh   -> gpurun_out/valu_share.txt
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out/valu"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
rm -rf "$OUT/p1"
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/p1" -o p -- python3 ${VALU_CMD:-"$REPO/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-extra} > "$OUT/p1.log" 2>&1
python3 - "$OUT/p1" <<'PY' | tee "$REPO/gpurun_out/valu_share.txt"
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
# last dispatch group = the timed step: take, per kernel name, the dispatches of the second half
by = collections.OrderedDict()
for r in rows:
    by.setdefault((r["Dispatch_Id"], r["Kernel_Name"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted({int(k[0]) for k in by})
half = ids[len(ids) // 2]
agg = collections.OrderedDict()
for (did, name), c in by.items():
    if int(did) < half or "SQ_INSTS_MFMA" not in c or c["SQ_INSTS_MFMA"] == 0:
        continue
    a = agg.setdefault(name[:70], collections.Counter())
    for k, v in c.items():
        a[k] += v
    a["n"] += 1
print(f"{'kernel':70s} {'n':>3s} {'MFMA inst':>10s} {'other VALU':>11s} {'VALU/MFMA':>9s} {'pipe busy%':>10s} {'valu issue%':>11s}")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"]):
    mf, va = a["SQ_INSTS_MFMA"], a["SQ_INSTS_VALU"] - a["SQ_INSTS_MFMA"]
    simd_cycles = a["GRBM_GUI_ACTIVE"] / 8 * 1024          # GRBM sums the 8 XCDs; 1024 SIMDs
    print(f"{name:70s} {a['n']:3d} {mf:10.3g} {va:11.3g} {va / mf:9.2f} {100 * a['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:10.1f} {100 * 4 * va / simd_cycles:11.1f}")
PY
