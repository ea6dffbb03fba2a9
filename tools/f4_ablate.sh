#!/bin/bash
# Where does the time of the F(4x4,3x3) kernel go?  Builds variants of libvatl_hip.so with one component of csrc/winograd_f4.hip removed (-DF4_ABL=bits: 1 no filter
# loads, 2 no LDS reads, 4 no staging DMA, 8 no stage barrier, 16 no output transform / write-out, 32 no wait for the first stage; WRONG results; F4_BITS="16 32 48" selects the variants) next to the product library.  Build container:  tools/f4_ablate.sh build
# GPU box: tools/f4_ablate.sh run [layers]
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; PKG="$REPO/vatl4pose-wacv2024_amd"
if [ "${1:-}" = build ]; then
  for b in ${F4_BITS:-1 2 4 8 3 15}; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$REPO/include" -I"$PKG/csrc" -DF4_ABL=$b -c "$PKG/csrc/winograd_f4.hip" -o /tmp/f4_abl_$b.o || exit 1
    objs=$(ls "$PKG"/build/*.o | grep -v winograd_f4.o)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$PKG/vatl_hip/libvatl_hip_f4abl$b.so" $objs /tmp/f4_abl_$b.o || exit 1
  done
  ls -la "$PKG"/vatl_hip/*.so
else
  shift
  L="${1:-l3.c2,hr.b128}"
  echo "as is"; python3 "$REPO/tools/f4_bench.py" --layers "$L" 2>&1 | grep "F(4x4)" | sed 's/.*F(4x4)/F(4x4)/' | cut -c1-40
  for b in ${F4_BITS:-1 2 4 8 3 15}; do
    echo "F4_ABL=$b"; VATL_HIP_LIB="$PKG/vatl_hip/libvatl_hip_f4abl$b.so" python3 "$REPO/tools/f4_bench.py" --layers "$L" 2>&1 | grep "F(4x4)" | sed 's/.*F(4x4)/F(4x4)/' | cut -c1-40
  done
fi
