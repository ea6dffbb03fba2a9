// Parameter guard of the inference plans (alphapose/models/hip_engine.py): one launch folds every parameter / buffer of a model into a
// 64-bit checksum per tensor.  A plan bakes packed weights and folded BatchNorm statistics in; it is keyed on the tensors' addresses and
// version counters, and an in-place write through `.data` (`p.data.copy_(w)`, the idiom of torch-1.12-era code such as
// alphapose/models/layers/dcn/deform_conv.py:232,255 and of hand-written checkpoint loaders around ActiveLearning.py:217) bumps no
// counter.  The checksums, launched in front of every plan call and compared with the ones taken when the plan was built, catch it.
//
// HBM-bound: every 32-bit word is read once (136 MB for SimplePose-R50, mostly resident in the Infinity Cache between calls).
// sum_i (w_i + 1) * (K + 2 i) over the ring of 64-bit integers, K odd: every word has its own ODD multiplier, so a change of any single
// word changes the sum with certainty (an odd multiplier is a bijection of the ring), two swapped words a, b at i, j move it by
// 2 (a - b)(i - j) != 0, and integer addition commutes, so blocks add their partial sums with one atomic each and the result does not
// depend on the order in which they finish.
#include "common.h"

namespace vatl {

constexpr long long kSumBlockWords = 16384;                   // 64 KB per block

__device__ __forceinline__ unsigned long long fold_word(unsigned w, long long i) {
    return ((unsigned long long)w + 1ull) * (0x9E3779B97F4A7C15ull + 2ull * (unsigned long long)i);
}

// table rows: {pointer, 32-bit words, first block} (int64 each); out: one 64-bit sum per row, zeroed by the caller
__global__ __launch_bounds__(256) void checksum_multi_kernel(const long long* __restrict__ table, int n_tensors, unsigned long long* __restrict__ out) {
    __shared__ int st;
    __shared__ unsigned long long part[4];
    if (threadIdx.x == 0) {
        int lo = 0, hi = n_tensors - 1;
        const long long b = blockIdx.x;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[3 * (long long)mid + 2] <= b) lo = mid; else hi = mid - 1;
        }
        st = lo;
    }
    __syncthreads();
    const long long* row = table + 3 * (long long)st;
    const unsigned* __restrict__ p = reinterpret_cast<const unsigned*>(row[0]);
    const long long n = row[1];
    const long long w0 = ((long long)blockIdx.x - row[2]) * kSumBlockWords;
    const long long w1 = w0 + kSumBlockWords < n ? w0 + kSumBlockWords : n;
    unsigned long long acc = 0;
    long long done = w0;
    if ((row[0] & 15) == 0) {                                 // w0 is a multiple of 4
        const long long q1 = w1 >> 2;
        for (long long q = (w0 >> 2) + threadIdx.x; q < q1; q += 256) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + 4 * q);
            acc += fold_word(v.x, 4 * q) + fold_word(v.y, 4 * q + 1) + fold_word(v.z, 4 * q + 2) + fold_word(v.w, 4 * q + 3);
        }
        done = q1 << 2;
    }
    for (long long i = done + threadIdx.x; i < w1; i += 256) acc += fold_word(p[i], i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)acc, o, 64), hi = __shfl_xor((unsigned)(acc >> 32), o, 64);
        acc += ((unsigned long long)hi << 32) | lo;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out + st, part[0] + part[1] + part[2] + part[3]);
}

}  // namespace vatl

using namespace vatl;

extern "C" int64_t vatl_checksum_block_words(void) { return kSumBlockWords; }

extern "C" int vatl_checksum_multi(const int64_t* table_dev, int n_tensors, int64_t total_blocks, uint64_t* out, void* stream) {
    if (n_tensors <= 0 || total_blocks <= 0) return 0;
    if (!table_dev || !out) return fail(VATL_EINVAL, "checksum_multi: null table / output");
    if (total_blocks > 0x7fffffffLL) return fail(VATL_EINVAL, "checksum_multi: %lld blocks", (long long)total_blocks);
    hipError_t e = hipMemsetAsync(out, 0, sizeof(uint64_t) * (size_t)n_tensors, (hipStream_t)stream);
    if (e != hipSuccess) return fail(VATL_ELAUNCH, "checksum_multi: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(checksum_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(table_dev), n_tensors, reinterpret_cast<unsigned long long*>(out));
    return check_launch("checksum_multi");
}
