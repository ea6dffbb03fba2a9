#!/bin/bash
# Round evidence in one GPU-box visit: rocprofv3 kernel statistics of the headline bench, of the fine-tune steps (configs 3 / 5)
# and of the HRNet pass (config 4), the HBM-traffic PMC passes of the bench, and the config_bench lines.
#   tools/profile_round.sh <tag>        -> gpurun_out/<tag>/...   (copy the summaries you want judged into profiles/)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r03}"
OUT="$REPO/gpurun_out/$TAG"
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
prof() {   # name, script args...
  local name="$1"; shift
  rm -rf "$OUT/$name"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o p -- python3 "$@" > "$OUT/$name.log" 2>&1
  grep '^{' "$OUT/$name.log" | tail -3 | cut -c1-400
  cp "$OUT/$name/p_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv" 2>/dev/null
}
echo "== bench kernel stats"; prof bench "$REPO/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-extra
python3 "$REPO/tools/layer_report.py" "$OUT/bench/p_kernel_trace.csv" > "$OUT/layer_report.txt" 2>&1; tail -3 "$OUT/layer_report.txt"
echo "== train step (config 3), single stream so that per-kernel durations add up to the step"
prof train "$REPO/tools/train_bench.py" --steps 7 --warmup 0 --single-stream
python3 "$REPO/tools/gap_report.py" "$OUT/train/p_kernel_trace.csv" > "$OUT/train_gap_report.json" 2>&1; cat "$OUT/train_gap_report.json" | cut -c1-400
echo "== train step (config 3), default (weight gradients on the side stream)"; prof train_overlap "$REPO/tools/train_bench.py" --steps 7 --warmup 0
python3 "$REPO/tools/gap_report.py" "$OUT/train_overlap/p_kernel_trace.csv" > "$OUT/train_overlap_gap_report.json" 2>&1
echo "== HRNet pass (config 4)"; prof hrnet "$REPO/tools/config_bench.py" --only cfg4
echo "== FastPose-R152 step (config 5)"; prof cfg5 "$REPO/tools/config_bench.py" --only cfg5 --single-stream
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $c"
  rm -rf "$OUT/pmc_$c"
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o r1 -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra > "$OUT/pmc_$c.log" 2>&1
  tail -1 "$OUT/pmc_$c.log" | cut -c1-200
done
find "$OUT" -name "*kernel_trace*" -size +20M -delete
find "$OUT" -name "*counter_collection*" -size +40M -delete
cd "$REPO"
echo "== scorer roofline"; timeout 600 python3 tools/scorer_bench.py 2>&1 | grep '^{' > "$OUT/scorer_roofline.jsonl"; wc -l "$OUT/scorer_roofline.jsonl"
echo "== bench (full line)"; timeout 900 python3 bench.py --steps 20 --warmup 3 2>&1 | grep '^{' | tail -1 > "$OUT/bench.json"; cut -c1-300 "$OUT/bench.json"
echo "== config_bench"; timeout 900 python3 tools/config_bench.py 2>&1 | grep '^{' | tee "$OUT/config_bench.jsonl"
python3 tools/pmc_summary.py "$OUT" "$OUT/pmc_summary.json" 2>&1 | tail -3
du -sh "$OUT"
