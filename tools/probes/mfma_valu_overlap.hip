// Does a SIMD's matrix pipe run one wave's MFMAs while its vector ALU runs another wave's (or the same wave's) ordinary vector instructions?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_valu_overlap tools/probes/mfma_valu_overlap.hip && /tmp/mfma_valu_overlap
// Each block = WAVES waves on one CU (WAVES / 4 per SIMD); every wave runs ITER rounds of [MF independent MFMAs (f32 16x16x4: 32 cycles each, or 32x32x2: 64)] and
// [VA dependent-free v_fma_f32].  Reported: cycles per round per SIMD against MF x passes and VA x 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MF, int VA, bool BIG, bool SPLIT>
__global__ __launch_bounds__(512) void probe(float* out, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[8];
    f32x16 accb[2];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 2; ++i) for (int e = 0; e < 16; ++e) accb[i][e] = 0.f;
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = (float)(lane + i);
    const float a = 1.0f + lane * 1e-3f, b = 0.5f;
    // SPLIT: even waves of a SIMD pair only do MFMAs, odd waves only vector instructions (different waves); otherwise every wave does both
    const bool do_m = !SPLIT || ((wave >> 2) & 1) == 0, do_v = !SPLIT || ((wave >> 2) & 1) == 1;
    for (int it = 0; it < iters; ++it) {
        if (do_m) {
#pragma unroll
            for (int k = 0; k < MF; ++k) {
                if constexpr (BIG) accb[k & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accb[k & 1], 0, 0, 0);
                else acc[k & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[k & 7], 0, 0, 0);
            }
        }
        if (do_v) {
#pragma unroll
            for (int k = 0; k < VA; ++k) v[k & 7] = __builtin_fmaf(v[k & 7], 1.0001f, 0.5f);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + v[i];
    s += accb[0][0] + accb[1][5];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MF, int VA, bool BIG, bool SPLIT>
static void run(const char* name, int waves) {
    float* out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    probe<MF, VA, BIG, SPLIT><<<256, waves * 64>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MF, VA, BIG, SPLIT><<<256, waves * 64>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double wps = waves / 4.0;                                  // waves per SIMD
    const double per_round_us = ms * 1e3 / iters;
    printf("%-58s %d waves/SIMD: %8.3f us per round per wave-set\n", name, (int)wps, per_round_us);
    hipFree(out);
}

int main() {
    // clocks warm-up
    run<64, 0, false, false>("warm-up", 8);
    printf("-- 16x16x4 f32 MFMA (8 passes = 32 cycles), 64 per round = 2048 cycles alone; v_fma 4 cycles each\n");
    run<64, 0, false, false>("64 MFMA, 0 vector, every wave", 4);
    run<64, 0, false, false>("64 MFMA, 0 vector, every wave", 8);
    run<0, 256, false, false>("0 MFMA, 256 vector (1024 cycles), every wave", 4);
    run<0, 256, false, false>("0 MFMA, 256 vector, every wave", 8);
    run<64, 256, false, false>("64 MFMA + 256 vector in the SAME wave", 4);
    run<64, 256, false, false>("64 MFMA + 256 vector in every wave", 8);
    run<64, 256, false, true>("64 MFMA in one wave, 256 vector in the OTHER wave of the SIMD", 8);
    run<64, 512, false, true>("64 MFMA in one wave, 512 vector in the OTHER wave of the SIMD", 8);
    printf("-- 32x32x2 f32 MFMA (16 passes = 64 cycles), 32 per round = 2048 cycles alone\n");
    run<32, 0, true, false>("32 MFMA, 0 vector, every wave", 4);
    run<32, 256, true, false>("32 MFMA + 256 vector in the SAME wave", 4);
    run<32, 256, true, true>("32 MFMA in one wave, 256 vector in the OTHER wave of the SIMD", 8);
    run<32, 512, true, true>("32 MFMA in one wave, 512 vector in the OTHER wave of the SIMD", 8);
    return 0;
}
