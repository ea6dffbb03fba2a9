cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -f gpurun_out/parity_report.jsonl
VATL_WGRAD_STREAM=${WS:-1} timeout 900 python -m pytest tests/test_gpu_train.py -q -k "every_block" 2>&1 | grep -E "^E  .*AssertionError|passed|failed" | cut -c1-3000
grep "block_check\|block_grad" gpurun_out/parity_report.jsonl | cut -c1-220 | grep -v "e-0[6789]" | head -40
