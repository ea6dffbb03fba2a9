// Third group of scorers: the heat-map criteria beside HP
//   MPE / Margin  compute_mpe / compute_margin    ActiveLearning.py:762-788  (skimage.feature.peak_local_max(min_distance=5,
//                                                  num_peaks=5) + scipy softmax / entropy)
//   Entropy       compute_entropy                  ActiveLearning.py:790-796  (scipy.stats.entropy of the flattened map)
// One block per (item, joint) plane; the plane lives in LDS.
#include "common.h"

namespace vatl {

// Block-wide arg-max over (value desc, index asc); -inf entries never win.  Returns the flat index or -1.
__device__ __forceinline__ int block_argmax(const float* v, int n, float* sval, int* sidx, float& best_val) {
    float bv = -INFINITY; int bi = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < n; i += 256) {
        const float x = v[i];
        if (x > bv) { bv = x; bi = i; }                        // strided scan keeps the lowest index among equals per thread
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sval[threadIdx.x >> 6] = bv; sidx[threadIdx.x >> 6] = bi; }
    __syncthreads();
    bv = sval[0]; bi = sidx[0];
#pragma unroll
    for (int w = 1; w < 4; ++w)
        if (sval[w] > bv || (sval[w] == bv && sidx[w] < bi)) { bv = sval[w]; bi = sidx[w]; }
    __syncthreads();
    best_val = bv;
    return bv == -INFINITY ? -1 : bi;
}

// peak_local_max(plane, min_distance = D, num_peaks = 5): (2D+1)^2 maximum filter with replicated edges, strictly
// above the plane minimum, D-wide border excluded, greedy spacing (Chebyshev distance < D rejected) in descending
// intensity / ascending index order, first five kept.  Also the per-plane MPE and Margin terms.
__global__ __launch_bounds__(256) void peaks5_kernel(const float* __restrict__ hm, float* __restrict__ peak_val, int32_t* __restrict__ peak_idx,
                                                     int32_t* __restrict__ npeaks, float* __restrict__ mpe, float* __restrict__ margin,
                                                     int H, int W, int D) {
    extern __shared__ float sm[];
    float* img = sm;                 // plane, later the candidate values (-inf = not a candidate)
    float* tmp = sm + H * W;         // row-filtered plane
    __shared__ float sval[4]; __shared__ int sidx[4];
    const int HW = H * W;
    const float* src = hm + (long long)blockIdx.x * HW;
    float mn = INFINITY;
    for (int i = threadIdx.x; i < HW; i += 256) { const float v = src[i]; img[i] = v; mn = fminf(mn, v); }
    mn = -wave_max(-mn);
    if ((threadIdx.x & 63) == 0) sval[threadIdx.x >> 6] = mn;
    __syncthreads();
    mn = fminf(fminf(sval[0], sval[1]), fminf(sval[2], sval[3]));
    __syncthreads();
    // separable (2D+1)^2 maximum filter with replicated edges (= maximum over the in-range part of the window).
    // Row pass: each thread produces 4 consecutive outputs from one sliding window (4 + 2D reads instead of 4 (2D+1)).
    const float inv_w = 1.0f / (float)W;
    const int W4 = (W + 3) >> 2;
    const float inv_w4 = 1.0f / (float)W4;
    for (int q = threadIdx.x; q < H * W4; q += 256) {
        const int y = fast_div(q, inv_w4), x = 4 * (q - y * W4);
        const float* row = img + y * W;
        float core = -INFINITY;                                   // columns x+3-D .. x+D are common to all four windows
        for (int xx = max(x + 3 - D, 0); xx <= min(x + D, W - 1); ++xx) core = fmaxf(core, row[xx]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (x + e >= W) break;
            float m = core;
            for (int xx = max(x + e - D, 0); xx < max(x + 3 - D, 0); ++xx) m = fmaxf(m, row[xx]);          // left extras
            for (int xx = min(x + D, W - 1) + 1; xx <= min(x + e + D, W - 1); ++xx) m = fmaxf(m, row[xx]);   // right extras
            tmp[y * W + x + e] = m;
        }
    }
    __syncthreads();
    // Column pass, same sliding scheme over 4 consecutive rows of one column
    const int H4 = (H + 3) >> 2;
    for (int q = threadIdx.x; q < H4 * W; q += 256) {
        const int yb = fast_div(q, inv_w), x = q - yb * W;
        const int y = 4 * yb;
        float core = -INFINITY;
        for (int yy = max(y + 3 - D, 0); yy <= min(y + D, H - 1); ++yy) core = fmaxf(core, tmp[yy * W + x]);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int yo = y + e;
            if (yo >= H) break;
            float m = core;
            for (int yy = max(yo - D, 0); yy < max(y + 3 - D, 0); ++yy) m = fmaxf(m, tmp[yy * W + x]);
            for (int yy = min(y + D, H - 1) + 1; yy <= min(yo + D, H - 1); ++yy) m = fmaxf(m, tmp[yy * W + x]);
            const float v = img[yo * W + x];
            const bool inside = yo >= D && yo < H - D && x >= D && x < W - D;
            img[yo * W + x] = (inside && v == m && v > mn) ? v : -INFINITY;     // each thread rewrites only what it alone reads
        }
    }
    __syncthreads();
    // Candidates are few (a handful per plane): compact them into a list and let the greedy selection run over the
    // list; a plateau-ridden plane with more than CAP candidates falls back to arg-max passes over the whole plane.
    constexpr int CAP = 256;
    __shared__ float cval[CAP]; __shared__ int cidx[CAP]; __shared__ int ccount;
    if (threadIdx.x == 0) ccount = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float v = img[i];
        if (v != -INFINITY) { const int slot = atomicAdd(&ccount, 1); if (slot < CAP) { cval[slot] = v; cidx[slot] = i; } }
    }
    __syncthreads();
    const int nc = ccount;
    float pv[5]; int pi[5]; int n = 0;
    if (nc <= CAP) {
        for (int k = 0; k < 5; ++k) {                            // block_argmax orders by (value desc, flat index asc): slot order is irrelevant
            float bv = -INFINITY; int bi = 0x7FFFFFFF;
            for (int t = threadIdx.x; t < nc; t += 256) {
                const float x = cval[t]; const int xi = cidx[t];
                if (x > bv || (x == bv && xi < bi)) { bv = x; bi = xi; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if ((threadIdx.x & 63) == 0) { sval[threadIdx.x >> 6] = bv; sidx[threadIdx.x >> 6] = bi; }
            __syncthreads();
            bv = sval[0]; bi = sidx[0];
#pragma unroll
            for (int w = 1; w < 4; ++w)
                if (sval[w] > bv || (sval[w] == bv && sidx[w] < bi)) { bv = sval[w]; bi = sidx[w]; }
            __syncthreads();
            if (bv == -INFINITY) break;
            pv[n] = bv; pi[n] = bi; ++n;
            const int by = bi / W, bx = bi - by * W;
            for (int t = threadIdx.x; t < nc; t += 256) {        // reject everything at Chebyshev distance < D (itself included)
                const int yy = fast_div(cidx[t], inv_w), xx = cidx[t] - yy * W;
                if (abs(yy - by) < D && abs(xx - bx) < D) cval[t] = -INFINITY;
            }
            __syncthreads();
        }
    } else {
        for (int k = 0; k < 5; ++k) {
            float bv;
            const int bi = block_argmax(img, HW, sval, sidx, bv);
            if (bi < 0) break;
            pv[n] = bv; pi[n] = bi; ++n;
            const int by = bi / W, bx = bi - by * W;
            const int side = 2 * D - 1;
            for (int t = threadIdx.x; t < side * side; t += 256) {
                const int yy = by - (D - 1) + t / side, xx = bx - (D - 1) + t % side;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) img[yy * W + xx] = -INFINITY;
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0) {
        const long long o = (long long)blockIdx.x;
        npeaks[o] = n;
        for (int k = 0; k < 5; ++k) { peak_val[o * 5 + k] = k < n ? pv[k] : 0.f; peak_idx[o * 5 + k] = k < n ? pi[k] : -1; }
        float e = 0.f;
        if (n > 0) {                                            // entropy(softmax(peaks)), float32 like scipy on a float32 array
            float ex[5], s = 0.f;
            for (int k = 0; k < n; ++k) { ex[k] = expf(pv[k] - pv[0]); s += ex[k]; }
            float q[5], qs = 0.f;
            for (int k = 0; k < n; ++k) { q[k] = ex[k] / s; qs += q[k]; }
            for (int k = 0; k < n; ++k) { const float p = q[k] / qs; e += p > 0.f ? -p * logf(p) : 0.f; }
        }
        mpe[o] = e;
        margin[o] = n > 1 ? fabsf(pv[0] - pv[1]) : 0.f;
    }
}

// The same function for the shipped heat-map size (64 x 48) and min_distance (5: ActiveLearning.py:773,784) with ONE WAVE per
// plane and no LDS / block barriers: lane = row, the row's 48 values live in registers.  Row pass of the 11 x 11 maximum
// filter: in registers (clipped window = replicated edges).  Column pass: the window [y-5, y+5] is the union of a backward run
// (y-5..y) and a forward run (y..y+5), each built by doubling from lane shuffles (6 per column, out-of-range rows contribute
// -inf).  Candidates stay in registers; the greedy selection is five rounds of a per-lane scan + wave arg-max (value
// descending, flat index ascending — the order of the block kernel) + in-register suppression.  The block kernel above spent
// its time in 17 barriers and an LDS compaction per plane (1300 us per 4096 items = 0.65 TB/s; this kernel: 437 us = 1.96 TB/s).
template <int W>
__global__ __launch_bounds__(256) void peaks5_wave_kernel(const float* __restrict__ hm, float* __restrict__ peak_val, int32_t* __restrict__ peak_idx,
                                                          int32_t* __restrict__ npeaks, float* __restrict__ mpe, float* __restrict__ margin,
                                                          long long planes) {
    constexpr int D = 5, H = 64;
    static_assert(W % 4 == 0 && W > 2 * D && W <= 64, "row of W floats per lane");
    // Round 4: the 11 x 11 maximum filter as two in-register passes of the doubling scheme (windows of 2, 4, 8, then 8 + 4 overlapping = 11:
    // 4 max operations per pixel and pass instead of 10), the COLUMN pass on a transposed copy of the plane — written row-wise to a
    // wave-private LDS tile (pitch W + 4 floats: conflict-free both ways), read column-wise (lane = column), filtered, written back and
    // read row-wise again — instead of six lane shuffles and eight max / select operations per pixel.  Round 3's kernel was bound by its
    // ~1200 vector and ~290 LDS-permute instructions per plane (2.1 TB/s); this one issues ~600 + ~150.  Same window maxima (max is
    // associative; fmaxf treats a NaN as missing in any order), same candidates, same greedy selection.
    constexpr int P = W + 4;
    extern __shared__ __attribute__((aligned(16))) float tile_all[];
    const int lane = threadIdx.x & 63;
    float* tile = tile_all + (threadIdx.x >> 6) * (H * P);
    const long long plane = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const float* src = hm + plane * (H * W) + lane * W;
    float v[W];
#pragma unroll
    for (int k = 0; k < W / 4; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(src + 4 * k);
        v[4 * k] = t[0]; v[4 * k + 1] = t[1]; v[4 * k + 2] = t[2]; v[4 * k + 3] = t[3];
        *reinterpret_cast<f32x4*>(tile + lane * P + 4 * k) = t;
    }
    float mn = v[0];
#pragma unroll
    for (int x = 1; x < W; ++x) mn = fminf(mn, v[x]);
    mn = -wave_max(-mn);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // column pass (lane = column x < W): c[y] = max over rows y - 5 .. y + 5 inside the plane
    if (lane < W) {
        float c[H];
#pragma unroll
        for (int y = 0; y < H; ++y) c[y] = tile[y * P + lane];
        float a2[H], a4[H], a8[H];                         // maxima of the windows [y, y + 2), [y, y + 4), [y, y + 8) clipped at the last row
#pragma unroll
        for (int y = 0; y < H; ++y) a2[y] = y + 1 < H ? fmaxf(c[y], c[y + 1]) : c[y];
#pragma unroll
        for (int y = 0; y < H; ++y) a4[y] = y + 2 < H ? fmaxf(a2[y], a2[y + 2]) : a2[y];
#pragma unroll
        for (int y = 0; y < H; ++y) a8[y] = y + 4 < H ? fmaxf(a4[y], a4[y + 4]) : a4[y];
#pragma unroll
        for (int y = 0; y < H; ++y) {
            // rows y - 5 .. y + 5 = [s, s + 8) U [y + 2, y + 6) with s = y - 5; at the top the window starts at row 0 and is shorter
            float m;
            if (y >= D) m = y + 2 < H ? fmaxf(a8[y - D], a4[y + 2]) : a8[y - D];
            else if (y + D + 1 >= 8) m = fmaxf(a8[0], a4[y + 2]);                  // [0, 8) U [y + 2, y + 6)   (y = 2 .. 4)
            else m = fmaxf(a4[0], a4[y + 2]);                                       // [0, 4) U [y + 2, y + 6)   (y = 0, 1: windows of 6 / 7 rows)
            tile[y * P + lane] = m;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // row pass (lane = row y) on the column maxima
    float h[W];
#pragma unroll
    for (int k = 0; k < W / 4; ++k) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(tile + lane * P + 4 * k);
        h[4 * k] = t[0]; h[4 * k + 1] = t[1]; h[4 * k + 2] = t[2]; h[4 * k + 3] = t[3];
    }
    float cand[W];
    {
        float b2[W], b4[W], b8[W];
#pragma unroll
        for (int x = 0; x < W; ++x) b2[x] = x + 1 < W ? fmaxf(h[x], h[x + 1]) : h[x];
#pragma unroll
        for (int x = 0; x < W; ++x) b4[x] = x + 2 < W ? fmaxf(b2[x], b2[x + 2]) : b2[x];
#pragma unroll
        for (int x = 0; x < W; ++x) b8[x] = x + 4 < W ? fmaxf(b4[x], b4[x + 4]) : b4[x];
        const bool row_in = lane >= D && lane < H - D;
#pragma unroll
        for (int x = D; x < W - D; ++x) {                  // candidates exist inside the border only: there the window is never clipped on the left
            const float m = fmaxf(b8[x - D], b4[x + 2]);
            cand[x] = (row_in && v[x] == m && v[x] > mn) ? v[x] : -INFINITY;
        }
#pragma unroll
        for (int x = 0; x < D; ++x) { cand[x] = -INFINITY; cand[W - 1 - x] = -INFINITY; }
    }
    float pv[5]; int pi[5]; int n = 0;
    // Greedy selection.  A plane has few candidates (maxima of 11 x 11 windows: typically 10 - 25): they are compacted into a list of at most
    // 64 — one per lane, in LDS where the tile was — and the five rounds run on ONE value per lane (wave arg-max + one distance test) instead
    // of re-scanning 48 registers per lane and round, which was two thirds of this kernel's vector instructions.  Planes with more than 64
    // candidates (plateaus of equal values) take the register scan below.  Same order either way: value descending, flat index ascending.
    int mine = 0;
#pragma unroll
    for (int x = D; x < W - D; ++x) mine += cand[x] > -INFINITY ? 1 : 0;
    int incl = mine;                                       // inclusive prefix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
    const int total = __shfl(incl, 63, 64);
    if (total <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                   // every lane has read its row of the tile
        int pos = incl - mine;
#pragma unroll
        for (int x = D; x < W - D; ++x)
            if (cand[x] > -INFINITY) { tile[pos] = cand[x]; reinterpret_cast<int*>(tile)[64 + pos] = lane * W + x; ++pos; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float cv = lane < total ? tile[lane] : -INFINITY;
        const int ci = lane < total ? reinterpret_cast<const int*>(tile)[64 + lane] : 0x7FFFFFFF;
        const int cy = ci / W, cx = ci - cy * W;
#pragma unroll 1
        for (int k = 0; k < 5; ++k) {
            float bv = cv; int bi = cv > -INFINITY ? ci : 0x7FFFFFFF;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
            }
            if (bv == -INFINITY) break;
            pv[n] = bv; pi[n] = bi; ++n;
            const int by = bi / W, bx = bi - by * W;
            if (abs(cy - by) < D && abs(cx - bx) < D) cv = -INFINITY;
        }
    } else {
#pragma unroll 1
    for (int k = 0; k < 5; ++k) {
        float bv = -INFINITY; int bi = 0x7FFFFFFF;
#pragma unroll
        for (int x = 0; x < W; ++x)
            if (cand[x] > bv) { bv = cand[x]; bi = lane * W + x; }          // ascending x: the first maximum of the row
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64); const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if (bv == -INFINITY) break;
        pv[n] = bv; pi[n] = bi; ++n;
        const int by = bi / W, bx = bi - by * W;
        if (abs(lane - by) < D) {
#pragma unroll
            for (int x = 0; x < W; ++x)
                if (abs(x - bx) < D) cand[x] = -INFINITY;
        }
    }
    }
    if (lane == 0) {
        npeaks[plane] = n;
        for (int k = 0; k < 5; ++k) { peak_val[plane * 5 + k] = k < n ? pv[k] : 0.f; peak_idx[plane * 5 + k] = k < n ? pi[k] : -1; }
        float e = 0.f;
        if (n > 0) {                                            // entropy(softmax(peaks)), float32 like scipy on a float32 array
            float ex[5], s = 0.f;
            for (int k = 0; k < n; ++k) { ex[k] = expf(pv[k] - pv[0]); s += ex[k]; }
            float q[5], qs = 0.f;
            for (int k = 0; k < n; ++k) { q[k] = ex[k] / s; qs += q[k]; }
            for (int k = 0; k < n; ++k) { const float p = q[k] / qs; e += p > 0.f ? -p * logf(p) : 0.f; }
        }
        mpe[plane] = e;
        margin[plane] = n > 1 ? fabsf(pv[0] - pv[1]) : 0.f;
    }
}

// scipy.stats.entropy(plane.flatten()): p = h / sum(h); sum of entr(p) with entr(p) = -p ln p (p > 0), 0 (p == 0),
// -inf (p < 0); a zero sum gives nan like numpy's 0/0 and x/0.
__global__ __launch_bounds__(256) void plane_entropy_kernel(const float* __restrict__ hm, float* __restrict__ out, int HW) {
    const float* gsrc = hm + (long long)blockIdx.x * HW;
    extern __shared__ __attribute__((aligned(16))) float plane[];   // read from HBM once; each thread re-reads only what it wrote
    __shared__ double sh[4];
    double s = 0.0;
    if ((HW & 3) == 0) {                               // 16-byte loads
        for (int i = threadIdx.x; i < (HW >> 2); i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(gsrc + 4 * i);
            *reinterpret_cast<f32x4*>(plane + 4 * i) = v;
            s += ((double)v[0] + (double)v[1]) + ((double)v[2] + (double)v[3]);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += 256) { const float v = gsrc[i]; plane[i] = v; s += (double)v; }
    }
    const float* src = plane;
    const int step = (HW & 3) == 0 ? 4 : 1;            // pass 2 walks the elements this thread staged
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    const float total = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
    __syncthreads();
    double e = 0.0;
    for (int i0 = threadIdx.x * step; i0 < HW; i0 += 256 * step) {
        for (int i = i0; i < i0 + step; ++i) {
            const float p = src[i] / total;
            float t;
            if (p > 0.f) t = -p * logf(p);
            else if (p == 0.f) t = 0.f;
            else if (p < 0.f) t = -INFINITY;
            else t = p;                                         // nan
            e += (double)t;
        }
    }
    e = wave_sum(e);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (float)(sh[0] + sh[1] + sh[2] + sh[3]);
}

// Same arithmetic with one WAVE per plane and the plane held in registers (NV float4 per lane, all loads in flight at once, no
// LDS, no block barrier): used when the plane is exactly 64 * NV float4 (64x48 -> NV = 12, 96x72 -> NV = 27).
template <int NV>
__global__ __launch_bounds__(256) void plane_entropy_wave_kernel(const float* __restrict__ hm, float* __restrict__ out, int planes) {
    const int lane = threadIdx.x & 63;
    const long long plane = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const f32x4* src = reinterpret_cast<const f32x4*>(hm + plane * (64LL * NV * 4)) + lane;
    f32x4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = src[k * 64];
    double s = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k) s += ((double)v[k][0] + (double)v[k][1]) + ((double)v[k][2] + (double)v[k][3]);
    const float total = (float)wave_sum(s);
    double e = 0.0;
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float p = v[k][c] / total;
            float t;
            if (p > 0.f) t = -p * logf(p);
            else if (p == 0.f) t = 0.f;
            else if (p < 0.f) t = -INFINITY;
            else t = p;                                         // nan
            e += (double)t;
        }
    e = wave_sum(e);
    if (lane == 0) out[plane] = (float)e;
}

// compute_OKS (al_metric.py:42-69): one thread per item, float64 like the numpy original.
__global__ void oks_kernel(const float* __restrict__ pred, const double* __restrict__ gt, const double* __restrict__ bbox_xywh,
                           double* __restrict__ out, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double sig[17] = {.026, .025, .025, .035, .035, .079, .079, .072, .072, .062, .062, .107, .107, .087, .087, .089, .089};
    const double bx = bbox_xywh[4 * i], by = bbox_xywh[4 * i + 1], bw = bbox_xywh[4 * i + 2], bh = bbox_xywh[4 * i + 3];
    const double area = bw * bh + 2.220446049250313e-16;                  // np.spacing(1)
    const float* d = pred + (long long)i * 51;
    const double* g = gt + (long long)i * 51;
    bool any_vis = false;
    for (int k = 0; k < 17; ++k) any_vis |= g[3 * k + 2] > 0.0;
    double acc = 0.0; int cnt = 0;
    for (int k = 0; k < 17; ++k) {
        const double xd = (double)d[3 * k], yd = (double)d[3 * k + 1];
        double dx, dy;
        if (any_vis) {
            if (!(g[3 * k + 2] > 0.0)) continue;
            dx = xd - g[3 * k]; dy = yd - g[3 * k + 1];
        } else {
            dx = fmax(0.0, (bx - bw) - xd) + fmax(0.0, xd - (bx + 2 * bw));
            dy = fmax(0.0, (by - bh) - yd) + fmax(0.0, yd - (by + 2 * bh));
        }
        const double var = (sig[k] * 2) * (sig[k] * 2);
        acc += exp(-((dx * dx + dy * dy) / var / area * 0.5));
        ++cnt;
    }
    out[i] = acc / (double)cnt;
}

}  // namespace vatl

using namespace vatl;

extern "C" int vatl_oks(const float* pred_kpts, const double* gt_kpts, const double* bbox_xywh, double* out, int N, void* stream) {
    if (N <= 0) return 0;
    if (!pred_kpts || !gt_kpts || !bbox_xywh || !out) return fail(VATL_EINVAL, "oks: null pointer");
    hipLaunchKernelGGL(oks_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, (hipStream_t)stream, pred_kpts, gt_kpts, bbox_xywh, out, N);
    return check_launch("oks");
}

extern "C" int vatl_peaks5(const float* hm, float* peak_val, int32_t* peak_idx, int32_t* npeaks, float* mpe, float* margin,
                           int N, int J, int H, int W, int min_distance, void* stream) {
    if (N <= 0) return 0;
    if (!hm || !peak_val || !peak_idx || !npeaks || !mpe || !margin) return fail(VATL_EINVAL, "peaks5: null pointer");
    if (min_distance < 1 || H <= 2 * min_distance || W <= 2 * min_distance) return fail(VATL_EINVAL, "peaks5: %dx%d plane too small for min_distance %d", H, W, min_distance);
    if (H == 64 && W == 48 && min_distance == 5 && (((uintptr_t)hm) & 15) == 0) {      // the shipped heat-map size: one wave per plane
        const long long planes = (long long)N * J;
        hipLaunchKernelGGL(peaks5_wave_kernel<48>, dim3((unsigned)cdiv(planes, 4)), dim3(256), 4 * 64 * (48 + 4) * sizeof(float), (hipStream_t)stream, hm, peak_val,
                           peak_idx, npeaks, mpe, margin, planes);
        return check_launch("peaks5");
    }
    const size_t smem = 2 * (size_t)H * W * sizeof(float);
    if (smem > 60 * 1024) return fail(VATL_EINVAL, "peaks5: heat-map %dx%d too large for the LDS tile", H, W);
    hipLaunchKernelGGL(peaks5_kernel, dim3((unsigned)(N * J)), dim3(256), smem, (hipStream_t)stream, hm, peak_val, peak_idx, npeaks, mpe, margin, H, W, min_distance);
    return check_launch("peaks5");
}

extern "C" int vatl_plane_entropy(const float* hm, float* out, int N, int J, int H, int W, void* stream) {
    if (N <= 0) return 0;
    if (!hm || !out) return fail(VATL_EINVAL, "plane_entropy: null pointer");
    const size_t smem = (size_t)H * W * sizeof(float);
    if (smem > 60 * 1024) return fail(VATL_EINVAL, "plane_entropy: heat-map %dx%d too large for the LDS tile", H, W);
    const long long planes = (long long)N * J;
    const bool aligned = (((uintptr_t)hm) & 15) == 0;
    if (aligned && H * W == 64 * 12 * 4) hipLaunchKernelGGL(plane_entropy_wave_kernel<12>, dim3(cdiv(planes, 4)), dim3(256), 0, (hipStream_t)stream, hm, out, (int)planes);
    else if (aligned && H * W == 64 * 27 * 4) hipLaunchKernelGGL(plane_entropy_wave_kernel<27>, dim3(cdiv(planes, 4)), dim3(256), 0, (hipStream_t)stream, hm, out, (int)planes);
    else hipLaunchKernelGGL(plane_entropy_kernel, dim3((unsigned)(N * J)), dim3(256), smem, (hipStream_t)stream, hm, out, H * W);
    return check_launch("plane_entropy");
}
