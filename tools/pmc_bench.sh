#!/bin/bash
# HBM traffic of one bench step (FETCH_SIZE / WRITE_SIZE passes) -> gpurun_out/<tag>/pmc_summary.json : tools/pmc_bench.sh <tag>
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-pmc}"
OUT="$REPO/gpurun_out/$TAG"
mkdir -p "$OUT"; export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf "$OUT/pmc_$c"
  timeout 900 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc_$c" -o r1 -- python3 "$REPO/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-extra > "$OUT/pmc_$c.log" 2>&1
done
cd "$REPO"
python3 tools/pmc_summary.py "$OUT" "$OUT/pmc_summary.json" 2>&1 | tail -1 | cut -c1-200
python3 - "$OUT/pmc_summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["kernels"]
for k, v in d.items():
    if "winograd_kernel" in k:
        print(k, round(v["read_bytes_per_step"] / 1e9, 2), "GB read", round(v["write_bytes_per_step"] / 1e9, 2), "GB written")
PY
find "$OUT" -name "*counter_collection*" -size +40M -delete; find "$OUT" -name "*kernel_trace*" -size +20M -delete
