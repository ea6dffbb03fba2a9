// Heat-map scorers: each reads the (N,J,H,W) fp32 heat-maps once (HBM-bound).
//   decode      heatmap_to_coord_simple        alphapose/utils/transforms.py:550-583
//   thc         compute_thc + neighbour rule   active_learning/ActiveLearning.py:345-363, 747-760
//   local-peak  localpeak_mean                 active_learning/local_peak.py:5-22
//   wpu         compute_hybrid + AE + MSE      active_learning/Whole_body_AE/*, ActiveLearning.py:364-386
#include "common.h"

namespace vatl {

// --------------------------------------------------------------------------
// decode: one 256-thread block per (item, joint)
// --------------------------------------------------------------------------
// np.argmax ordering: larger value first, equal values by lower index, and NaN is the maximum (the FIRST NaN wins)
__device__ __forceinline__ void argmax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i) || (ov != ov && (v == v || oi < i))) { v = ov; i = oi; }
}

// quarter-pixel shift + inverse crop affine + stores of one plane's result (one thread)
__device__ __forceinline__ void decode_finish(const float* __restrict__ src, const float* __restrict__ bbox, float* __restrict__ coords,
                                              float* __restrict__ maxvals, int32_t* __restrict__ idx_out, long long plane, int item,
                                              float best, int bidx, int H, int W, int cs = 2, int ms = 1, int mo = 0) {
    // cs / ms / mo: element stride of a plane's coordinate pair / score and the score's offset — (2, 1, 0) for separate (N,J,2) + (N,J)
    // arrays, (3, 3, 2) with coords == maxvals for the interleaved (N,J,3) key-point rows of vatl_decode_pose
    int px = bidx % W, py = bidx / W;
    if (!(best > 0.f)) { px = 0; py = 0; }                      // pred_mask: maxval <= 0 zeroes the coords
    float u = (float)px, v = (float)py;
    if (1 < px && px < W - 1 && 1 < py && py < H - 1) {
        const float dx = src[py * W + px + 1] - src[py * W + px - 1];
        const float dy = src[(py + 1) * W + px] - src[(py - 1) * W + px];
        u += (dx > 0.f ? 0.25f : (dx < 0.f ? -0.25f : 0.f));
        v += (dy > 0.f ? 0.25f : (dy < 0.f ? -0.25f : 0.f));
    }
    // inverse crop affine: control points rounded to float32 like the reference's
    // np.float32 src/dst arrays, transform itself in float64 (cv2.getAffineTransform)
    const double xmin = bbox[item * 4 + 0], ymin = bbox[item * 4 + 1];
    const double xmax = bbox[item * 4 + 2], ymax = bbox[item * 4 + 3];
    const double bw = xmax - xmin, bh = ymax - ymin;
    const double cx = xmin + bw * 0.5, cy = ymin + bh * 0.5;
    const float cx32 = (float)cx, cy32 = (float)cy;
    const float top32 = (float)(cy + bw * -0.5);
    const float d32 = cy32 - top32;
    const double g = (double)d32 / (W * 0.5);
    coords[plane * cs + 0] = (float)((double)cx32 + ((double)u - W * 0.5) * g);
    coords[plane * cs + 1] = (float)((double)cy32 + ((double)v - H * 0.5) * g);
    maxvals[plane * ms + mo] = best;
    if (idx_out) idx_out[plane] = bidx;
}

__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ hm, const float* __restrict__ bbox,
                                                     float* __restrict__ coords, float* __restrict__ maxvals,
                                                     int32_t* __restrict__ idx_out, int J, int H, int W, int cs, int ms, int mo) {
    const int item = blockIdx.x / J;
    const int HW = H * W;
    const float* src = hm + (long long)blockIdx.x * HW;
    const int tid = threadIdx.x;

    // first maximum in row-major order: strict '>' inside a thread (indices ascend),
    // (value, lower index) ordering between threads
    float best = -INFINITY;
    int bidx = 0x7fffffff;
    if ((HW & 3) == 0) {
        const int n4 = HW >> 2;
        for (int q = tid; q < n4; q += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[e] > best || bidx == 0x7fffffff || (v[e] != v[e] && best == best)) { best = v[e]; bidx = 4 * q + e; }
        }
    } else {
        for (int q = tid; q < HW; q += 256) {
            const float v = src[q];
            if (v > best || bidx == 0x7fffffff || (v != v && best == best)) { best = v; bidx = q; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        argmax_merge(best, bidx, ov, oi);
    }
    __shared__ float sv[4];
    __shared__ int si[4];
    if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bidx; }
    __syncthreads();
    if (tid != 0) return;
    for (int w = 1; w < 4; ++w) argmax_merge(best, bidx, sv[w], si[w]);

    decode_finish(src, bbox, coords, maxvals, idx_out, (long long)blockIdx.x, item, best, bidx, H, W, cs, ms, mo);
}

// One WAVE per plane, the plane in registers (NV float4 per lane, every load in flight at once, no LDS, no block barrier): used
// when the plane is exactly 64 * NV float4 (64x48 -> NV = 12, 96x72 -> NV = 27).  Same ordering rules, same finish.
template <int NV>
__global__ __launch_bounds__(256) void decode_wave_kernel(const float* __restrict__ hm, const float* __restrict__ bbox,
                                                          float* __restrict__ coords, float* __restrict__ maxvals,
                                                          int32_t* __restrict__ idx_out, int planes, int J, int H, int W, int cs, int ms, int mo) {
    const int lane = threadIdx.x & 63;
    const long long plane = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const float* src = hm + plane * (64LL * NV * 4);
    const f32x4* src4 = reinterpret_cast<const f32x4*>(src) + lane;
    f32x4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = src4[k * 64];
    float best = -INFINITY;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (v[k][e] > best || bidx == 0x7fffffff || (v[k][e] != v[k][e] && best == best)) { best = v[k][e]; bidx = 4 * (k * 64 + lane) + e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bidx, o, 64);
        argmax_merge(best, bidx, ov, oi);
    }
    if (lane == 0) decode_finish(src, bbox, coords, maxvals, idx_out, plane, (int)(plane / J), best, bidx, H, W, cs, ms, mo);
}

// Per-item pose scores from the interleaved key-point rows: HP = -np.sum(scores) (ActiveLearning.py:329-330) and the json "score" =
// np.mean(scores) + 1.25 np.max(scores) (:314), with NumPy's float32 pairwise summation order (8 running sums over the leading multiple
// of 8, combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), the tail added in order; halves of a run longer than 128 summed separately) —
// HP is bit-identical to the reference's float32 np.sum, not merely close.  One thread per item.
__device__ float np_pairwise_sum(const float* a, int n, int stride) {
    if (n < 8) {
        float r = 0.f;
        for (int i = 0; i < n; ++i) r += a[i * stride];
        return r;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = a[j * stride];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] += a[(i + j) * stride];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i * stride];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2, stride) + np_pairwise_sum(a + (long long)n2 * stride, n - n2, stride);
}

__global__ void pose_scores_kernel(const float* __restrict__ kpts, float* __restrict__ hp, double* __restrict__ pose_score, int N, int J) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const float* sc = kpts + (long long)i * J * 3 + 2;
    const float sum = np_pairwise_sum(sc, J, 3);
    float mx = sc[0];
    for (int j = 1; j < J; ++j) {                              // np.max: NaN propagates
        const float v = sc[j * 3];
        if (v > mx || v != v) mx = (mx != mx) ? mx : v;
    }
    if (hp) hp[i] = -sum;
    // float(np.mean(s) + 1.25 * np.max(s)) under the reference's pinned numpy==1.23.5 (pyproject.toml:35): np.mean of float32 scores is a float32
    // scalar (pairwise sum / J, correctly rounded); `1.25 * np.float32` is a python float times a NumPy SCALAR, which numpy 1.x promotes to float64,
    // so the product (exact in float64) and the sum (one rounding, in float64) are doubles and the json "score" is that double.  (NumPy >= 2 keeps the
    // whole expression in float32; round 5 followed that and was up to 1 ulp of float32 away from the reference's file.  alphapose/utils/bbox.py
    // follows the same 1.23 promotion for _center_scale_to_box.)
    if (pose_score) pose_score[i] = (double)(sum / (float)J) + 1.25 * (double)mx;
}

// --------------------------------------------------------------------------
// THC: one block per pair of heat-map stacks (J*H*W floats each)
// --------------------------------------------------------------------------
template <int NORM>
__global__ __launch_bounds__(256) void thc_pairs_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        long long sa, long long sb, float* __restrict__ out, int J, int n) {
    const float* pa = a + (long long)blockIdx.x * sa;
    const float* pb = b + (long long)blockIdx.x * sb;
    const int tid = threadIdx.x;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    const int n4 = n >> 2;
    for (int q = tid; q < n4; q += 256) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(pa + 4 * q);
        const f32x4 y = *reinterpret_cast<const f32x4*>(pb + 4 * q);
        const float d0 = x[0] - y[0], d1 = x[1] - y[1], d2 = x[2] - y[2], d3 = x[3] - y[3];
        if (NORM == 1) { acc0 += fabsf(d0); acc1 += fabsf(d1); acc2 += fabsf(d2); acc3 += fabsf(d3); }
        else           { acc0 += d0 * d0;   acc1 += d1 * d1;   acc2 += d2 * d2;   acc3 += d3 * d3; }
    }
    for (int q = 4 * n4 + tid; q < n; q += 256) {
        const float d = pa[q] - pb[q];
        acc0 += NORM == 1 ? fabsf(d) : d * d;
    }
    double s = wave_sum((double)acc0 + (double)acc1 + (double)acc2 + (double)acc3);
    __shared__ double part[4];
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) out[blockIdx.x] = (float)((part[0] + part[1] + part[2] + part[3]) / (double)J);
}

__global__ void thc_combine_kernel(const float* __restrict__ pair, const uint8_t* __restrict__ is_prev,
                                   const uint8_t* __restrict__ is_next, float* __restrict__ thc, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const bool hp = is_prev[i] != 0 && i > 0, hn = is_next[i] != 0 && i < N - 1;
    float t = 0.f;
    if (hp) t += pair[i - 1];
    if (hn) t += pair[i];
    if (hp != hn) t *= 2.f;                                     // exactly one neighbour: doubled
    thc[i] = t;
}

// --------------------------------------------------------------------------
// local peaks: one block per (item, joint) plane staged in LDS with a zero halo; per-plane (sum, count) of the kept
// peaks go to a workspace and a second kernel forms the per-item mean in joint order (deterministic)
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void localpeak_plane_kernel(const float* __restrict__ hm, double* __restrict__ ws, int32_t* __restrict__ count_out,
                                                              int H, int W, float order) {
    extern __shared__ __attribute__((aligned(16))) float tile[];        // (H+2) x (W+2), zero halo = mode='constant', cval=0
    __shared__ float wmax[4];
    __shared__ int wcnt[4];
    __shared__ double wsum[4];
    const int tid = threadIdx.x;
    const int HW = H * W, PW = W + 2;
    const float inv_w = 1.0f / (float)W;
    const float* src = hm + (long long)blockIdx.x * HW;
    // interior: 16-byte global loads (W % 4 == 0: a float4 never straddles rows); halo: zeros
    const int n4 = HW >> 2, W4 = W >> 2;
    const float inv_w4 = 1.0f / (float)W4;
    for (int q = tid; q < n4; q += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * q);
        const int y = fast_div(q, inv_w4), x = 4 * (q - y * W4);
        float* d = tile + (y + 1) * PW + x + 1;
        d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3];
    }
    for (int q = tid; q < PW; q += 256) { tile[q] = 0.f; tile[(H + 1) * PW + q] = 0.f; }
    for (int q = tid; q < H; q += 256) { tile[(q + 1) * PW] = 0.f; tile[(q + 1) * PW + W + 1] = 0.f; }
    __syncthreads();
    // pass 1: local maxima (pixel equal to its 3x3 zero-padded max) and the largest of them.  Each thread owns 1x4
    // strips: 18 LDS reads serve four pixels.  Peak flags are kept in a per-thread bit set for pass 2.
    float pmax = -INFINITY;
    unsigned long long flags = 0ull;                           // strip k of this thread, pixel e  ->  bit 4k + e   (n4 <= 16 * 256)
    int k = 0;
    for (int q = tid; q < n4; q += 256, ++k) {
        const int y = fast_div(q, inv_w4), x = 4 * (q - y * W4);
        const float* c = tile + (y + 1) * PW + (x + 1);
        float up[6], mid[6], dn[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) { up[e] = c[-PW - 1 + e]; mid[e] = c[-1 + e]; dn[e] = c[PW - 1 + e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = mid[e + 1];
            float m = fmaxf(fmaxf(up[e], up[e + 1]), up[e + 2]);
            m = fmaxf(m, fmaxf(mid[e], mid[e + 2]));
            m = fmaxf(m, fmaxf(fmaxf(dn[e], dn[e + 1]), dn[e + 2]));
            if (v >= m) { pmax = fmaxf(pmax, v); flags |= 1ull << (4 * k + e); }
        }
    }
    pmax = wave_max(pmax);
    if ((tid & 63) == 0) wmax[tid >> 6] = pmax;
    __syncthreads();
    pmax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    // pass 2: keep peaks >= order * largest peak
    const float thr = pmax * order;
    float s = 0.f;
    int cnt = 0;
    if (pmax > -INFINITY && flags) {
        k = 0;
        for (int q = tid; q < n4; q += 256, ++k) {
            const unsigned f = (unsigned)(flags >> (4 * k)) & 15u;
            if (!f) continue;
            const int y = fast_div(q, inv_w4), x = 4 * (q - y * W4);
            const float* c = tile + (y + 1) * PW + (x + 1);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((f >> e) & 1u) { const float v = c[e]; if (v >= thr) { s += v; ++cnt; } }
        }
    }
    (void)inv_w;
    const double ds = wave_sum((double)s);
    cnt = wave_sum(cnt);
    if ((tid & 63) == 0) { wsum[tid >> 6] = ds; wcnt[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        const int c = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        ws[2 * (long long)blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        ws[2 * (long long)blockIdx.x + 1] = (double)c;
        if (count_out) count_out[blockIdx.x] = c;
    }
}

// Register-tile variant for W % 4 == 0 (every shipped heat-map size): the plane never touches LDS.  A thread owns a 4-column x
// RPT-row patch: RPT + 2 float4 row loads (the two halo rows come out of L1/L2), the 3x3 maximum is a vertical max3 in registers
// followed by a horizontal max3 whose outer columns come from the neighbouring lanes (one shuffle each way per row).  Lanes
// are laid out [row group][float4 column] with whole row groups per wave, so a neighbour is always lane +- 1 of the same wave.
// Zero halo (scipy mode='constant', cval=0): rows / columns outside the plane read as 0.
template <int RPT>
__global__ __launch_bounds__(1024) void localpeak_plane_reg_kernel(const float* __restrict__ hm, double* __restrict__ ws, int32_t* __restrict__ count_out,
                                                                   int H, int W, float order) {
    __shared__ float wmax[16];
    __shared__ int wcnt[16];
    __shared__ double wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int W4 = W >> 2, gpw = 64 / W4;                     // float4 columns per row, row groups per wave
    const int g = lane / W4, c4 = lane - g * W4;
    const bool on = g < gpw;
    const int row0 = (wave * gpw + g) * RPT;                  // first row of this thread's patch
    const float* src = hm + (long long)blockIdx.x * H * W + 4 * c4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    f32x4 v[RPT + 2];
#pragma unroll
    for (int r = 0; r < RPT + 2; ++r) {
        const int y = row0 - 1 + r;
        v[r] = (on && y >= 0 && y < H) ? *reinterpret_cast<const f32x4*>(src + (long long)y * W) : zero;
    }
    float pmax = -INFINITY;
    unsigned long long flags = 0ull;                          // row r, column e of the patch -> bit 4 r + e
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        f32x4 vm;
#pragma unroll
        for (int e = 0; e < 4; ++e) vm[e] = fmaxf(fmaxf(v[r][e], v[r + 1][e]), v[r + 2][e]);
        float left = __shfl_up(vm[3], 1, 64), right = __shfl_down(vm[0], 1, 64);
        if (c4 == 0) left = 0.f;
        if (c4 == W4 - 1) right = 0.f;
        const float L[4] = {fmaxf(fmaxf(left, vm[0]), vm[1]), fmaxf(fmaxf(vm[0], vm[1]), vm[2]), fmaxf(fmaxf(vm[1], vm[2]), vm[3]),
                            fmaxf(fmaxf(vm[2], vm[3]), right)};
        const bool live = on && row0 + r < H;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float c = v[r + 1][e];
            if (live && c >= L[e]) { pmax = fmaxf(pmax, c); flags |= 1ull << (4 * r + e); }
        }
    }
    pmax = wave_max(pmax);
    if (lane == 0) wmax[wave] = pmax;
    __syncthreads();
    pmax = wmax[0];
    for (int w = 1; w < nw; ++w) pmax = fmaxf(pmax, wmax[w]);
    // pass 2: keep peaks >= order * largest peak (values still in registers)
    const float thr = pmax * order;
    float s = 0.f;
    int cnt = 0;
    if (pmax > -INFINITY && flags) {
#pragma unroll
        for (int r = 0; r < RPT; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if ((flags >> (4 * r + e)) & 1ull) { const float c = v[r + 1][e]; if (c >= thr) { s += c; ++cnt; } }
    }
    const double ds = wave_sum((double)s);
    cnt = wave_sum(cnt);
    if (lane == 0) { wsum[wave] = ds; wcnt[wave] = cnt; }
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        int c = 0;
        for (int w = 0; w < nw; ++w) { t += wsum[w]; c += wcnt[w]; }
        ws[2 * (long long)blockIdx.x] = t;
        ws[2 * (long long)blockIdx.x + 1] = (double)c;
        if (count_out) count_out[blockIdx.x] = c;
    }
}

// any width (the per-item API accepts arbitrary planes): scalar loads, one pixel per thread step
__global__ __launch_bounds__(256) void localpeak_plane_generic_kernel(const float* __restrict__ hm, double* __restrict__ ws, int32_t* __restrict__ count_out,
                                                              int H, int W, float order) {
    extern __shared__ __attribute__((aligned(16))) float tile[];        // (H+2) x (W+2)
    __shared__ float wmax[4];
    __shared__ int wcnt[4];
    __shared__ double wsum[4];
    const int tid = threadIdx.x;
    const int HW = H * W, PW = W + 2, PN = (H + 2) * PW;
    const float* src = hm + (long long)blockIdx.x * HW;
    for (int q = tid; q < PN; q += 256) {
        const int y = q / PW - 1, x = q % PW - 1;
        tile[q] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? src[y * W + x] : 0.f;   // mode='constant', cval=0
    }
    __syncthreads();
    // pass 1: largest local maximum (pixel equal to its 3x3 zero-padded max); the peak flag is kept in the sign of a
    // per-thread bit set so that pass 2 does not redo the nine reads
    float pmax = -INFINITY;
    unsigned long long flags = 0ull;                           // element q = tid + 256 k  ->  bit k  (HW <= 64 * 256)
    int k = 0;
    for (int q = tid; q < HW; q += 256, ++k) {
        const int y = q / W, x = q - y * W;
        const float* c = tile + (y + 1) * PW + (x + 1);
        const float v = c[0];
        float m = fmaxf(fmaxf(c[-PW - 1], c[-PW]), c[-PW + 1]);
        m = fmaxf(m, fmaxf(c[-1], c[1]));
        m = fmaxf(m, fmaxf(fmaxf(c[PW - 1], c[PW]), c[PW + 1]));
        if (v >= m) { pmax = fmaxf(pmax, v); flags |= 1ull << k; }
    }
    pmax = wave_max(pmax);
    if ((tid & 63) == 0) wmax[tid >> 6] = pmax;
    __syncthreads();
    pmax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    // pass 2: keep peaks >= order * largest peak
    const float thr = pmax * order;
    float s = 0.f;
    int cnt = 0;
    if (pmax > -INFINITY) {
        k = 0;
        for (int q = tid; q < HW; q += 256, ++k) {
            if (!((flags >> k) & 1ull)) continue;
            const int y = q / W, x = q - y * W;
            const float v = tile[(y + 1) * PW + (x + 1)];
            if (v >= thr) { s += v; ++cnt; }
        }
    }
    const double ds = wave_sum((double)s);
    cnt = wave_sum(cnt);
    if ((tid & 63) == 0) { wsum[tid >> 6] = ds; wcnt[tid >> 6] = cnt; }
    __syncthreads();
    if (tid == 0) {
        const int c = wcnt[0] + wcnt[1] + wcnt[2] + wcnt[3];
        ws[2 * (long long)blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        ws[2 * (long long)blockIdx.x + 1] = (double)c;
        if (count_out) count_out[blockIdx.x] = c;
    }
}

__global__ void localpeak_finish_kernel(const double* __restrict__ ws, float* __restrict__ mean_out, int N, int J) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double tot = 0.0, cnt = 0.0;
    for (int j = 0; j < J; ++j) { tot += ws[2 * ((long long)i * J + j)]; cnt += ws[2 * ((long long)i * J + j) + 1]; }
    mean_out[i] = cnt > 0.0 ? (float)(tot / cnt) : __builtin_nanf("");
}

// --------------------------------------------------------------------------
// WPU: one wave per item; lane l owns feature / neuron l
// --------------------------------------------------------------------------
__device__ __forceinline__ float dense_lane(const float* __restrict__ Wt, const float* __restrict__ bias, int n_out, int n_in,
                                            float h, int lane) {
    // out[lane] = bias[lane] + sum_k W[lane][k] * h_k, h_k broadcast from lane k
    float acc = lane < n_out ? bias[lane] : 0.f;
    for (int k = 0; k < n_in; ++k) {
        const float hk = __shfl(h, k, 64);
        if (lane < n_out) acc = fmaf(Wt[lane * n_in + k], hk, acc);
    }
    return acc;
}

__global__ __launch_bounds__(256) void wpu_kernel(const float* __restrict__ kpts, const float* __restrict__ bbox,
                                                  const float* __restrict__ ae, int D, int z, int only38,
                                                  float* __restrict__ wpu, int32_t* __restrict__ status, int N) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= N) return;
    const float* kp = kpts + (long long)item * 51;
    // bbox_xyxy_to_xywh: h = ymax - ymin + 1 (alphapose/utils/bbox.py:96)
    const double height = (double)bbox[item * 4 + 3] - (double)bbox[item * 4 + 1] + 1.0;
    double sw = 0.0, sx = 0.0, sy = 0.0;
    for (int j = 0; j < 17; ++j) {
        const double s = kp[3 * j + 2];
        sw += s; sx += (double)kp[3 * j] * s; sy += (double)kp[3 * j + 1] * s;
    }
    int st = 0;
    if (!(height > 0.0)) st = 1;
    else if (!(sw > 0.0)) st = 2;
    if (st) {
        if (lane == 0) { wpu[item] = __builtin_nanf(""); if (status) status[item] = st; }
        return;
    }
    const double gx = sx / sw, gy = sy / sw;
    // 42-d feature: (x-gx)/h [17], (y-gy)/h [17], 8 joint-triangle angles
    double f = 0.0;
    if (lane < 17) f = ((double)kp[3 * lane] - gx) / height;
    else if (lane < 34) f = ((double)kp[3 * (lane - 17) + 1] - gy) / height;
    else if (lane < 42) {
        const int tri[8][3] = {{8, 6, 12}, {6, 8, 10}, {5, 7, 9}, {7, 5, 11}, {11, 12, 14}, {12, 11, 13}, {12, 14, 16}, {11, 13, 15}};
        const int t = lane - 34;
        const double x0 = kp[3 * tri[t][0]], y0 = kp[3 * tri[t][0] + 1];
        const double x1 = kp[3 * tri[t][1]], y1 = kp[3 * tri[t][1] + 1];
        const double x2 = kp[3 * tri[t][2]], y2 = kp[3 * tri[t][2] + 1];
        const double eps = 1e-6;
        const double m1 = (y1 - y0) / (x1 - x0 + eps);
        const double m2 = (y2 - y1) / (x2 - x1 + eps);
        f = atan(fabs((m1 - m2) / (1.0 + m1 * m2 + eps)));
    }
    // D == 38: the auto-encoder's declared width takes the 38-value subset (drop 3,4,20,21)
    float x0f = (float)f;
    if (D == 38) {
        const int srcl = lane < 3 ? lane : (lane < 18 ? lane + 2 : lane + 4);
        x0f = __shfl(x0f, srcl & 63, 64);
        if (lane >= 38) x0f = 0.f;
    }
    const int dims[5] = {D, 24, 12, 7, z};
    const float* w = ae;
    float h = x0f;
    for (int i = 0; i < 4; ++i) {                               // encoder: ReLU after all but the code layer
        const int ni = dims[i], no = dims[i + 1];
        h = dense_lane(w, w + no * ni, no, ni, h, lane);
        if (i < 3) h = fmaxf(h, 0.f);
        w += no * ni + no;
    }
    for (int i = 4; i > 0; --i) {                               // decoder: ReLU, Sigmoid on the last
        const int ni = dims[i], no = dims[i - 1];
        h = dense_lane(w, w + no * ni, no, ni, h, lane);
        h = i > 1 ? fmaxf(h, 0.f) : 1.f / (1.f + expf(-h));
        w += no * ni + no;
    }
    float d = 0.f;
    int cnt = D;
    if (lane < D) {
        bool use = true;
        if (only38) use = !(lane == 3 || lane == 4 || lane == 20 || lane == 21);
        d = use ? (h - x0f) * (h - x0f) : 0.f;
    }
    if (only38) cnt = D - 4;
    const float s = wave_sum(d);
    if (lane == 0) { wpu[item] = s / (float)cnt; if (status) status[item] = 0; }
}

}  // namespace vatl

using namespace vatl;

static int decode_launch(const float* hm, const float* bbox, float* coords, float* maxvals, int32_t* idx, int N, int J, int H, int W, int cs, int ms,
                         int mo, hipStream_t st) {
    const long long planes = (long long)N * J;
    const bool aligned = (((uintptr_t)hm) & 15) == 0 && planes <= 0x7FFFFFFF;
    if (aligned && H * W == 64 * 12 * 4)
        hipLaunchKernelGGL(decode_wave_kernel<12>, dim3(cdiv(planes, 4)), dim3(256), 0, st, hm, bbox, coords, maxvals, idx, (int)planes, J, H, W, cs, ms, mo);
    else if (aligned && H * W == 64 * 27 * 4)
        hipLaunchKernelGGL(decode_wave_kernel<27>, dim3(cdiv(planes, 4)), dim3(256), 0, st, hm, bbox, coords, maxvals, idx, (int)planes, J, H, W, cs, ms, mo);
    else
        hipLaunchKernelGGL(decode_kernel, dim3(N * J), dim3(256), 0, st, hm, bbox, coords, maxvals, idx, J, H, W, cs, ms, mo);
    return check_launch("decode_argmax_affine");
}

extern "C" int vatl_decode_argmax_affine(const float* hm, const float* bbox, float* coords, float* maxvals, int32_t* idx,
                                         int N, int J, int H, int W, void* stream) {
    if (N <= 0) return 0;
    if (!hm || !bbox || !coords || !maxvals) return fail(VATL_EINVAL, "decode_argmax_affine: null pointer");
    return decode_launch(hm, bbox, coords, maxvals, idx, N, J, H, W, 2, 1, 0, (hipStream_t)stream);
}

extern "C" int vatl_decode_pose(const float* hm, const float* bbox, float* kpts, int32_t* idx, float* hp, double* pose_score, int N, int J, int H, int W,
                                void* stream) {
    if (N <= 0) return 0;
    if (!hm || !bbox || !kpts) return fail(VATL_EINVAL, "decode_pose: null pointer");
    if (int rc = decode_launch(hm, bbox, kpts, kpts, idx, N, J, H, W, 3, 3, 2, (hipStream_t)stream)) return rc;
    if (hp || pose_score) {
        hipLaunchKernelGGL(pose_scores_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, kpts, hp, pose_score, N, J);
        return check_launch("pose_scores");
    }
    return 0;
}

extern "C" int vatl_thc_pairs(const float* a, const float* b, int64_t stride_a, int64_t stride_b, float* out,
                              int P, int J, int HW, int norm, void* stream) {
    if (norm != 1 && norm != 2) return fail(VATL_EINVAL, "thc_pairs: norm must be 1 (L1) or 2 (L2)");
    if (P <= 0) return 0;
    if (!a || !b || !out) return fail(VATL_EINVAL, "thc_pairs: null pointer");
    if (((uintptr_t)a | (uintptr_t)b) & 15 || (stride_a & 3) || (stride_b & 3))
        return fail(VATL_EINVAL, "thc_pairs: operands must be 16-byte aligned");
    if (norm == 1) hipLaunchKernelGGL(thc_pairs_kernel<1>, dim3(P), dim3(256), 0, (hipStream_t)stream, a, b, (long long)stride_a, (long long)stride_b, out, J, J * HW);
    else           hipLaunchKernelGGL(thc_pairs_kernel<2>, dim3(P), dim3(256), 0, (hipStream_t)stream, a, b, (long long)stride_a, (long long)stride_b, out, J, J * HW);
    return check_launch("thc_pairs");
}

extern "C" int vatl_thc_combine(const float* pair, const uint8_t* is_prev, const uint8_t* is_next, float* thc, int N, void* stream) {
    if (N <= 0) return 0;
    if (!is_prev || !is_next || !thc || (N > 1 && !pair)) return fail(VATL_EINVAL, "thc_combine: null pointer");
    hipLaunchKernelGGL(thc_combine_kernel, dim3(cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, pair, is_prev, is_next, thc, N);
    return check_launch("thc_combine");
}

extern "C" int vatl_localpeak_mean(const float* hm, float* mean, int32_t* count, double* workspace, int N, int J, int H, int W, float order,
                                   void* stream) {
    if (N <= 0) return 0;
    if (!hm || !mean || !workspace) return fail(VATL_EINVAL, "localpeak_mean: null pointer");
    const size_t smem = (size_t)(H + 2) * (W + 2) * sizeof(float);
    if (smem > 60 * 1024 || (long long)H * W > 64 * 256) return fail(VATL_EINVAL, "localpeak_mean: heat-map %dx%d too large for the LDS tile", H, W);
    const int W4 = W >> 2, gpw = W4 > 0 ? 64 / W4 : 0;
    const int reg_waves = gpw > 0 ? cdiv(H, 16 * gpw) : 1 << 30;     // register-tile kernel: 16 rows per thread, whole row groups per wave
    if (W & 3) hipLaunchKernelGGL(localpeak_plane_generic_kernel, dim3((unsigned)(N * J)), dim3(256), smem, (hipStream_t)stream, hm, workspace, count, H, W, order);
    else if (reg_waves <= 16 && (((uintptr_t)hm) & 15) == 0)
        hipLaunchKernelGGL(localpeak_plane_reg_kernel<16>, dim3((unsigned)(N * J)), dim3(64 * reg_waves), 0, (hipStream_t)stream, hm, workspace, count, H, W, order);
    else       hipLaunchKernelGGL(localpeak_plane_kernel, dim3((unsigned)(N * J)), dim3(256), smem, (hipStream_t)stream, hm, workspace, count, H, W, order);
    hipLaunchKernelGGL(localpeak_finish_kernel, dim3(cdiv(N, 128)), dim3(128), 0, (hipStream_t)stream, workspace, mean, N, J);
    return check_launch("localpeak_mean");
}

extern "C" int vatl_hybrid_ae_wpu(const float* kpts, const float* bbox, const float* ae, int D, int z, int only38,
                                  float* wpu, int32_t* status, int N, void* stream) {
    if (N <= 0) return 0;
    if (!kpts || !bbox || !ae || !wpu) return fail(VATL_EINVAL, "hybrid_ae_wpu: null pointer");
    if (D != 38 && D != 42) return fail(VATL_EINVAL, "hybrid_ae_wpu: D must be 38 or 42, got %d", D);
    if (z < 1 || z > 64) return fail(VATL_EINVAL, "hybrid_ae_wpu: code width %d out of range", z);
    if (only38 && D != 42) return fail(VATL_EINVAL, "hybrid_ae_wpu: only38 needs D == 42");
    if (N <= 0) return 0;
    hipLaunchKernelGGL(wpu_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, kpts, bbox, ae, D, z, only38, wpu, status, N);
    return check_launch("hybrid_ae_wpu");
}
