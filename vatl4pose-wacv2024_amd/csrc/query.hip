// Query selection on the embeddings — the step right after the scoring path:
//   influence / diversity  sum of cosine distances to every other item    ActiveLearning.py:467-476, 583-592
//                          (sklearn KNeighborsTransformer(mode='distance', metric='cosine', n_neighbors=n-1) + row sums)
//   core-set               k-center greedy with an uncertainty term        ActiveLearning.py:798-850
//                          (sklearn pairwise_distances(..., 'euclidean') + np.minimum + np.argmax)
// Embeddings are the fp32 (N, D) rows get_embedding returns; like the reference's float64 fvecs_matrix every
// accumulation is float64.  All HBM-bound: N*D*4 bytes per pass.
#include "common.h"

namespace vatl {

// inv_norm[i] = 1 / ||x_i||  (1 when the row is zero, like sklearn.preprocessing.normalize)
__global__ __launch_bounds__(256) void row_inv_norm_kernel(const float* __restrict__ x, double* __restrict__ inv_norm, long long n, int D) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    double s = 0.0;
    for (int d = lane; d < D; d += 64) { const double v = x[row * D + d]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) inv_norm[row] = s > 0.0 ? 1.0 / sqrt(s) : 1.0;
}

// s[d] = sum_i x[i][d] * inv_norm[i]   (one thread per column, rows in order: deterministic)
__global__ __launch_bounds__(256) void unit_colsum_kernel(const float* __restrict__ x, const double* __restrict__ inv_norm, double* __restrict__ s,
                                                          long long n, int D) {
    const int d = blockIdx.x * 256 + threadIdx.x;
    if (d >= D) return;
    double acc = 0.0;
    for (long long i = 0; i < n; ++i) acc += (double)x[i * D + d] * inv_norm[i];
    s[d] = acc;
}

// out[i] = sum_j (1 - cos(x_i, x_j)) = n - x^_i . s
__global__ __launch_bounds__(256) void cosine_rowsum_kernel(const float* __restrict__ x, const double* __restrict__ inv_norm, const double* __restrict__ s,
                                                            double* __restrict__ out, long long n, int D) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    double dot = 0.0;
    for (int d = lane; d < D; d += 64) dot += (double)x[row * D + d] * s[d];
    dot = wave_sum(dot);
    if (lane == 0) out[row] = (double)n - dot * inv_norm[row];
}

// min_dist[i] = min(min_dist[i], ||x_i - x_c||) for the centres c = centers[0..nc); `centers` lives on the device so
// that a freshly selected index feeds the next update without a host round trip.  first != 0: no previous value.
__global__ __launch_bounds__(256) void kcenter_update_kernel(const float* __restrict__ x, const int32_t* __restrict__ centers, int nc,
                                                             double* __restrict__ min_dist, long long n, int D, int first) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int lane = threadIdx.x & 63;
    double best = first ? INFINITY : min_dist[row];
    for (int k = 0; k < nc; ++k) {
        const long long c = centers[k];
        double s = 0.0;
        for (int d = lane; d < D; d += 64) { const double v = (double)x[row * D + d] - (double)x[c * D + d]; s += v * v; }
        s = wave_sum(s);
        best = fmin(best, sqrt(s));
    }
    if (lane == 0) min_dist[row] = best;
}

// sel[step] = first arg-max of a*min_dist + b*unc (np.argmax); then unc[sel] = 0.  One block.
__global__ __launch_bounds__(1024) void kcenter_pick_kernel(const double* __restrict__ min_dist, double* __restrict__ unc, double a, double b,
                                                            int32_t* __restrict__ sel, int step, long long n) {
    __shared__ double sv[16]; __shared__ long long si[16];
    double bv = -INFINITY; long long bi = 0x7FFFFFFFFFFFFFFFLL;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const double v = a * (min_dist ? min_dist[i] : 0.0) + b * (unc ? unc[i] : 0.0);
        if (v > bv) { bv = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ov = __shfl_xor(bv, o, 64); const long long oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (sv[w] > bv || (sv[w] == bv && si[w] < bi)) { bv = sv[w]; bi = si[w]; }
        if (bi == 0x7FFFFFFFFFFFFFFFLL) bi = 0;            // all -inf / nan: np.argmax returns 0
        sel[step] = (int32_t)bi;
        if (unc) unc[bi] = 0.0;
    }
}

}  // namespace vatl

using namespace vatl;

extern "C" int vatl_cosine_rowsum(const float* emb, int64_t n, int D, double* out, double* workspace, void* stream) {
    if (n <= 0) return 0;
    if (!emb || !out || !workspace || D <= 0) return fail(VATL_EINVAL, "cosine_rowsum: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    double* inv_norm = workspace;                // n doubles
    double* s = workspace + n;                   // D doubles
    hipLaunchKernelGGL(row_inv_norm_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, emb, inv_norm, (long long)n, D);
    hipLaunchKernelGGL(unit_colsum_kernel, dim3((unsigned)((D + 255) / 256)), dim3(256), 0, st, emb, inv_norm, s, (long long)n, D);
    hipLaunchKernelGGL(cosine_rowsum_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, st, emb, inv_norm, s, out, (long long)n, D);
    return check_launch("cosine_rowsum");
}

extern "C" int vatl_kcenter_update(const float* emb, int64_t n, int D, const int32_t* centers_dev, int n_centers, double* min_dist, int first,
                                   void* stream) {
    if (n <= 0 || n_centers <= 0) return 0;
    if (!emb || !centers_dev || !min_dist || D <= 0) return fail(VATL_EINVAL, "kcenter_update: bad arguments");
    hipLaunchKernelGGL(kcenter_update_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, emb, centers_dev, n_centers, min_dist,
                       (long long)n, D, first);
    return check_launch("kcenter_update");
}

extern "C" int vatl_kcenter_pick(const double* min_dist_or_null, double* unc_or_null, double a, double b, int32_t* selected_dev, int step,
                                 int64_t n, void* stream) {
    if (n <= 0) return fail(VATL_EINVAL, "kcenter_pick: empty pool");
    if (!selected_dev || step < 0 || (!min_dist_or_null && !unc_or_null)) return fail(VATL_EINVAL, "kcenter_pick: bad arguments");
    hipLaunchKernelGGL(kcenter_pick_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, min_dist_or_null, unc_or_null, a, b, selected_dev, step, (long long)n);
    return check_launch("kcenter_pick");
}
