// Shared helpers for libvatl_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstdint>

#include "vatl_hip.h"

namespace vatl {

// thread-local message behind vatl_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);

// Executed-MFMA-FLOP meter of the calling host thread (vatl_flop_meter_begin / vatl_flop_meter_end): every matrix-core launch adds
// what its grid really multiplies — 2 * padded rows * padded columns * padded reduction length, tile padding included — under
// kind 0 (direct sums: implicit GEMM, weight gradients) or kind 1 (Winograd transform-domain GEMMs).  Thread-local like the
// split-K scope: nothing global, off unless the thread asked for it.
void meter_add(int kind, double flops);
// ... and WHICH kernel family a launch took (vatl_flop_meter_routes): the tests assert that the configuration they pin really ran the
// route they name (two-half Winograd blocks, staging-address tables, the BatchNorm-backward epilogue, ...).  Same thread-local switch.
enum MeterRoute {
    kRouteIgemm = 0, kRouteIgemmBnBwd, kRouteIgemmDma, kRoutePersistent1x1, kRouteStreamK, kRouteRows, kRouteChain, kRouteStemPool, kRouteHalo,
    kRouteWino, kRouteWino2H, kRouteWinoBnBwd, kRouteWinoPersist, kRouteWinoC32, kRouteWgrad, kRouteWinoWgrad, kRouteWinoWgrad2H,
    kRouteWinoWgradTable, kRouteWinoF4, kRouteWinoF4BnBwd, kRouteCount
};
void meter_route(int route);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(VATL_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

// Raising a kernel's dynamic-LDS limit is a per-device setting: `done_mask` (one static per kernel instantiation) remembers
// the devices it has been applied on, so a process that drives several devices configures each of them once.
inline int ensure_dynamic_lds(const void* kern, int bytes, std::atomic<unsigned>& done_mask, const char* what) {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) return fail(VATL_ELAUNCH, "%s: no current device", what);
    const unsigned bit = 1u << (d & 31);
    if (done_mask.load(std::memory_order_acquire) & bit) return 0;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return fail(VATL_ELAUNCH, "hipFuncSetAttribute(%s): %s", what, hipGetErrorString(e));
    done_mask.fetch_or(bit, std::memory_order_release);
    return 0;
}

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// 64-lane butterfly reductions (wave = 64 on CDNA)
// Barrier for LDS hand-offs: waits for this wave's LDS operations only.  `__syncthreads()` is a workgroup release fence + barrier: with global STORES or LDS-DMA
// requests in flight hipcc emits `s_waitcnt vmcnt(0)` in front of it, and the vector memory counter retires in order — so a plain barrier in a write-out waits for
// the acknowledgement of every store issued before it and for every load / DMA request issued after those (skip-connection values and staging requests issued early
// on purpose).  Where the threads of a block exchange data through LDS only, this is the barrier to use; data that arrives by LDS-DMA needs its own counted
// `s_waitcnt vmcnt(N)` in front, as everywhere in csrc/.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// q / w for 0 <= q < 2^20, 1 <= w <= 4096 without an integer division: (q + 0.5) / w is at least 0.5 / w away from
// every integer, far more than the float32 rounding error of the product, so the truncation is exact.
__device__ __forceinline__ int fast_div(int q, float inv_w) { return (int)(((float)q + 0.5f) * inv_w); }

// Exact unsigned division by a launch constant without the ~40-instruction integer division: q / w = (mulhi(q, magic) + q) >> shift for
// 0 <= q < 2^31, 1 <= w < 2^31 (Granlund & Montgomery; shift = ceil(log2 w), magic = floor(2^32 (2^shift - w) / w) + 1).
struct FastDivU {
    unsigned magic, shift;
};
static inline FastDivU make_fastdiv(unsigned w) {
    FastDivU f{};
    unsigned sh = 0;
    while ((1ull << sh) < w) ++sh;
    f.shift = sh;
    f.magic = (unsigned)((((1ull << 32) * ((1ull << sh) - w)) / w) + 1);
    return f;
}
__device__ __forceinline__ int fdiv(int q, FastDivU f) { return (int)((__umulhi((unsigned)q, f.magic) + (unsigned)q) >> f.shift); }

}  // namespace vatl
