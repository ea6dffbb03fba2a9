#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel stats.  Logs -> gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
echo "== pytest -m gpu" ; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench" ; timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -5 | tee gpurun_out/bench.log
if [ "${1:-}" = "prof" ]; then
  echo "== rocprofv3 kernel stats"
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof.log 2>&1
  tail -3 gpurun_out/prof.log
  find gpurun_out/prof -name "*kernel_stats*" | head; f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -25 "$f"
  # keep only the small summaries
  find gpurun_out/prof -name "*kernel_trace*" -size +30M -delete
fi
