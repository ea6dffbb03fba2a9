#!/bin/bash
# round-3 probe: GPU tests + fine-tune step trace with idle-gap accounting
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$REPO"
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > "$OUT/gputest1.log"; cat "$OUT/gputest1.log"
cd /tmp
for mode in 1 0; do
  rm -rf "$OUT/train_ws$mode"
  VATL_WGRAD_STREAM=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/train_ws$mode" -o p -- python3 "$REPO/tools/train_bench.py" --steps 6 --warmup 3 > "$OUT/train_ws$mode.log" 2>&1
  grep '^{' "$OUT/train_ws$mode.log" | tail -1
  python3 "$REPO/tools/gap_report.py" "$OUT/train_ws$mode/p_kernel_trace.csv" | tee "$OUT/gap_ws$mode.json"
done
python3 "$REPO/tools/train_bench.py" --steps 30 --warmup 5 | tee "$OUT/train_plain.json"
find "$OUT" -name "*kernel_trace*" -size +20M -delete
