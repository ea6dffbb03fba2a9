cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_conv.py -q -x 2>&1 | tail -3
run() { VATL_PREPACK=$1 timeout 300 python tools/train_bench.py --steps 40 --warmup 8 --model $2 --batch $3 2>&1 | tail -1 | sed 's/.*ms_per_step": \([0-9.]*\).*/\1/'; }
for rep in 1 2; do for pp in 0 1; do
echo "rep $rep prepack $pp: simplepose $(run $pp simplepose 120)  fastpose $(run $pp fastpose 120)"
done; done
for pp in 0 1 0 1; do echo "cfg5 prepack $pp: $(VATL_PREPACK=$pp timeout 600 python tools/config_bench.py --only cfg5 2>&1 | tail -1 | cut -c60-140)"; done
