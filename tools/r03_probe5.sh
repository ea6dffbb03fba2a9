#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$REPO"
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -k "streamk" 2>&1 | tail -12
timeout 900 python -m pytest tests/test_gpu_train.py -x -q 2>&1 | tail -6
for sk in 1 0; do
  VATL_STREAMK=$sk timeout 300 python tools/train_bench.py --steps 30 --warmup 5
  VATL_STREAMK=$sk timeout 300 python tools/config_bench.py --only cfg5 2>&1 | grep '^{'
done
