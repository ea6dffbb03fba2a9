cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_conv.py -q -x 2>&1 | tail -2
timeout 600 python tools/config_bench.py --only cfg4 2>&1 | tail -1
cd /tmp; rm -rf $GRAFT_REPO_ROOT/gpurun_out/hprof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hprof -o h -- python3 $GRAFT_REPO_ROOT/tools/config_bench.py --only cfg4 > $GRAFT_REPO_ROOT/gpurun_out/hprof.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/hprof.log | cut -c1-200
find $GRAFT_REPO_ROOT/gpurun_out/hprof -name "*kernel_trace*" -size +40M -delete
