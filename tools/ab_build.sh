#!/bin/bash
# A/B helper: build libvatl_hip_base.so from a git revision (default HEAD) next to the working tree's libvatl_hip.so, so that one
# GPU-box visit can alternate the two libraries (VATL_HIP_LIB=<path> selects one; see vatl_hip/__init__.py).
#   tools/ab_build.sh [rev]
set -e
REV="${1:-HEAD}"
REPO="$(cd "$(dirname "$0")/.." && pwd)"
WT=/tmp/vatl_ab_wt
rm -rf "$WT"; git -C "$REPO" worktree prune; git -C "$REPO" worktree add -f --detach "$WT" "$REV" > /dev/null 2>&1
python3 "$WT/vatl4pose-wacv2024_amd/build.py" > /tmp/vatl_ab_build.log 2>&1 || { tail -20 /tmp/vatl_ab_build.log; exit 1; }
cp "$WT/vatl4pose-wacv2024_amd/vatl_hip/libvatl_hip.so" "$REPO/vatl4pose-wacv2024_amd/vatl_hip/libvatl_hip_base.so"
git -C "$REPO" worktree remove --force "$WT"
python3 "$REPO/vatl4pose-wacv2024_amd/build.py" > /tmp/vatl_ab_build2.log 2>&1 || { tail -20 /tmp/vatl_ab_build2.log; exit 1; }
ls -la "$REPO"/vatl4pose-wacv2024_amd/vatl_hip/*.so
