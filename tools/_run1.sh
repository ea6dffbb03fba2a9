cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { VATL_FUSE_BN_MINC=$1 timeout 300 python tools/train_bench.py --steps 40 --warmup 8 --model $2 --batch $3 2>&1 | tail -1 | sed 's/.*ms_per_step": \([0-9.]*\).*/\1/'; }
for rep in 1 2; do
for mc in 9999 256 128 64; do
echo "rep $rep MINC $mc: simplepose $(run $mc simplepose 120)  fastpose $(run $mc fastpose 120)  hrnet $(run $mc hrnet 64)"
done; done
