#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$REPO"
python -m pytest tests/test_gpu_train.py tests/test_gpu_conv.py tests/test_gpu_dist.py tests/test_gpu_api.py -x -q 2>&1 | tail -12 > "$OUT/gputest3.log"; cat "$OUT/gputest3.log"
python tools/train_bench.py --steps 30 --warmup 5 | tee "$OUT/train_plain3.json"
python tools/train_bench.py --steps 20 --warmup 5 --model fastpose | tee -a "$OUT/train_plain3.json"
