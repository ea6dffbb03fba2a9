#!/bin/bash
# Alternate libvatl_hip_base.so / libvatl_hip.so over a benchmark command on one GPU box (same box, interleaved runs: boxes differ by 2 - 4 %).
#   tools/ab_run.sh <rounds> <python script + args...>       prints the lines of every run prefixed with base / new
ROUNDS="$1"; shift
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 "$ROUNDS"); do
  VATL_HIP_LIB="$REPO/vatl4pose-wacv2024_amd/vatl_hip/libvatl_hip_base.so" python3 "$@" 2>&1 | grep -v amdgpu.ids | sed "s/^/base$i | /"
  python3 "$@" 2>&1 | grep -v amdgpu.ids | sed "s/^/new$i  | /"
done
