#!/bin/bash
# round-3 probe 2: full GPU test suite, scorer roofline (events + rocprofv3 kernel stats)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$REPO"
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > "$OUT/gputest2.log"; cat "$OUT/gputest2.log"
python tools/scorer_bench.py 2>&1 | grep '^{' | tee "$OUT/scorer_roofline.jsonl"
cd /tmp
rm -rf "$OUT/scorers"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/scorers" -o p -- python3 "$REPO/tools/scorer_bench.py" --iters 5 > "$OUT/scorers.log" 2>&1
cp "$OUT/scorers/p_kernel_stats.csv" "$OUT/scorer_kernel_stats.csv"
head -30 "$OUT/scorer_kernel_stats.csv" | cut -c1-200
find "$OUT" -name "*kernel_trace*" -size +20M -delete
