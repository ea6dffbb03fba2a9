cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_api.py -q -x -k "mpe" 2>&1 | tail -5
timeout 600 python tools/scorer_bench.py --only peaks5 2>&1 | tail -3
