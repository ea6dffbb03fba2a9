#!/bin/bash
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/r03"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$REPO"
python -m pytest tests/test_gpu_train.py tests/test_gpu_conv.py tests/test_gpu_dist.py tests/test_gpu_al.py -x -q 2>&1 | tail -12 > "$OUT/gputest4.log"; cat "$OUT/gputest4.log"
python tools/train_bench.py --steps 30 --warmup 5 | tee "$OUT/train_plain4.json"
VATL_PACK_PLAN=0 python tools/train_bench.py --steps 30 --warmup 5 | tee -a "$OUT/train_plain4.json"
python tools/config_bench.py --only cfg5 2>&1 | grep '^{' | tee -a "$OUT/train_plain4.json"
VATL_PACK_PLAN=0 python tools/config_bench.py --only cfg5 2>&1 | grep '^{' | tee -a "$OUT/train_plain4.json"
