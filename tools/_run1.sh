cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/al_eval_bench.py 2>&1 | tail -2
timeout 600 python tools/al_eval_bench.py --retrain 2>&1 | tail -2
