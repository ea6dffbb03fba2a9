#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, optional rocprof passes.  Logs -> gpurun_out/.
#   tools/gpu_round.sh [prof] [pmc] [notest] [nobench]
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
has() { [[ " $* " == *" $WANT "* ]]; }
ARGS=" $* "
rm -f gpurun_out/parity_report.jsonl
if [[ "$ARGS" != *" notest "* ]]; then
  echo "== pytest -m gpu" ; timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
  echo "== smoke" ; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
fi
if [[ "$ARGS" != *" nobench "* ]]; then
  echo "== bench" ; timeout 900 python bench.py --steps 5 --warmup 2 2>&1 | tail -5 | tee gpurun_out/bench.log
fi
if [[ "$ARGS" == *" prof "* ]]; then
  echo "== rocprofv3 kernel stats"
  rm -rf gpurun_out/prof
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/prof.log 2>&1
  tail -2 gpurun_out/prof.log
  f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-200
  find gpurun_out/prof -name "*kernel_trace*" -size +30M -delete
fi
if [[ "$ARGS" == *" pmc "* ]]; then
  # HBM traffic counters: one counter per pass, kernel-trace only (MI355X_MICROARCH.md §HBM)
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "== rocprofv3 --pmc $c"
    rm -rf gpurun_out/pmc_$c
    timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -o r1 -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > gpurun_out/pmc_$c.log 2>&1
    tail -1 gpurun_out/pmc_$c.log | cut -c1-300
    ls gpurun_out/pmc_$c | head
    find gpurun_out/pmc_$c -name "*kernel_trace*" -size +30M -delete
  done
fi
