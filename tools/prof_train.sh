#!/bin/bash
# rocprofv3 kernel statistics of one trainer's fine-tune step (single stream): tools/prof_train.sh <tag> [train_bench args...]
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-train}"; shift
OUT="$REPO/gpurun_out/$TAG"
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
rm -rf "$OUT/t"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/t" -o p -- python3 "$REPO/tools/train_bench.py" --steps 7 --warmup 0 --single-stream "$@" > "$OUT/t.log" 2>&1
grep '^{' "$OUT/t.log" | tail -1 | cut -c1-200
cp "$OUT/t/p_kernel_stats.csv" "$OUT/kernel_stats.csv"
python3 "$REPO/tools/gap_report.py" "$OUT/t/p_kernel_trace.csv" | tee "$OUT/gap_report.json" | cut -c1-600
python3 - "$OUT/t/p_kernel_trace.csv" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
# per (kernel, grid) totals over the last 5 steps' worth: just aggregate everything and divide by 7
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = (r["Kernel_Name"][:60], r.get("Grid_Size", r.get("Grid_Size_X", "")))
    agg[k][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); agg[k][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:45]:
    print(f"{v[0] / 7e3:9.1f} us/step  x{v[1] / 7:6.1f}  {k[0]}  grid {k[1]}")
PY
find "$OUT" -name "*kernel_trace*" -size +20M -delete
