timeout 600 python -m pytest tests/test_gpu_winograd.py -x -q -k "wgrad" 2>&1 | tail -3
for i in 1 2; do for h in 1 2; do echo "== halves $h"; timeout 300 python tools/wino_wgrad_bench.py --halves $h --iters 10 --layers l2.c2,l3.c2,l4.c2,r152.l3.c2,deconv1,deconv2,deconv3 2>&1 | grep -v amdgpu | cut -c1-24,60-140; done; done
