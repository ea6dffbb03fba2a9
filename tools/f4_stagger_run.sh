REPO="${GRAFT_REPO_ROOT:-/root/repo}"; PKG="$REPO/vatl4pose-wacv2024_amd"
for r in 1 2; do
echo "base"; python3 $REPO/tools/f4_bench.py --layers hr.b64,l1.c2,l2.c2,l3.c2,hr.b128 --iters 20 2>&1 | grep "F(4x4)" | sed 's/.*F(4x4)/F(4x4)/' | cut -c1-30
for b in 2 4 7; do echo "stagger $b"; VATL_HIP_LIB=$PKG/vatl_hip/libvatl_hip_f4stg$b.so python3 $REPO/tools/f4_bench.py --layers hr.b64,l1.c2,l2.c2,l3.c2,hr.b128 --iters 20 2>&1 | grep "F(4x4)" | sed 's/.*F(4x4)/F(4x4)/' | cut -c1-30; done
done
