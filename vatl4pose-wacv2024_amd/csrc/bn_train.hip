// Training-mode BatchNorm2d forward/backward on NHWC (M = N*H*W rows, C channels) and the
// other HBM-bound pieces of the fine-tune step (ActiveLearning.py:658-673 with model.train()):
//   forward   batch mean / biased variance -> per-channel (scale, bias); running-stat update
//             (momentum 0.1, unbiased variance into running_var); y = act(z*scale + bias (+ res))
//   backward  g = dy * [y > 0];  dbeta = sum g;  dgamma = sum g * xhat;
//             dz = gamma*invstd * (g - dbeta/M - xhat*dgamma/M)  = A[c]*g + B[c]*z + C[c]
//   max-pool 3x3/2 backward, per-channel column sums (conv bias gradient).
// Reductions are two-stage (per-block double partials, then a finalize kernel): deterministic.
#include "common.h"

namespace vatl {

constexpr int MAX_ROW_BLOCKS = 4096;     // partial sums per channel (the finalize kernels stride 64 lanes over them)

// Column reductions over NHWC rows: a 256-thread block owns a row range; thread t always handles the same four
// channels (float4 column t % C4 of a 256-column slab selected by blockIdx.y) and every 256/C4-th row, so global
// loads are 16-byte and fully coalesced; threads sharing a column are combined through LDS.  Partials are doubles:
// partial[(rb * C + c) * 2 + {0,1}].
struct RowSplit { long long rows_per_block; int nrb; };
static inline RowSplit row_split(long long M, int C) {
    // A block of col_reduce_kernel covers min(C/4, 256) float4 columns with 256 threads, i.e. `rstep` rows at a time.  Aim at
    // >= ~1024 blocks (small-M layers would otherwise run on a few dozen CUs with every thread walking hundreds of rows one
    // load at a time) while keeping >= 16 row steps per block, so that the per-block partials stay a few % of the traffic.
    const int c4 = C >> 2 > 0 ? C >> 2 : 1;
    const int cols = c4 < 256 ? c4 : 256, rstep = 256 / cols, gy = (c4 + 255) / 256;
    long long rpb = 256;                               // large M: ~256 rows per block, thousands of blocks
    const long long want = 1024 / gy > 0 ? 1024 / gy : 1;
    if ((M + rpb - 1) / rpb < want) {
        rpb = (M + want - 1) / want;
        const long long floor_rows = 16LL * rstep;
        if (rpb < floor_rows) rpb = floor_rows;
        rpb = (rpb + rstep - 1) / rstep * rstep;
    }
    long long nrb = (M + rpb - 1) / rpb;
    if (nrb > MAX_ROW_BLOCKS) { nrb = MAX_ROW_BLOCKS; rpb = (M + nrb - 1) / nrb; }
    if (nrb < 1) nrb = 1;
    return {rpb, (int)((M + rpb - 1) / rpb)};
}

// ReLU mask of a layer without a skip input, recomputed from z exactly as scale_bias_act_kernel produced y (same
// fmaf), so the backward passes need not read y
__device__ __forceinline__ bool relu_on(float z, float sc, float bi) { return fmaf(z, sc, bi) > 0.f; }

template <int MODE>   // 0: (sum z, sum z^2)   1: (sum g, sum g*xhat) with g = dy*[y>0]
__global__ __launch_bounds__(256) void col_reduce_kernel(const float* __restrict__ a, const float* __restrict__ y, const float* __restrict__ z,
                                                         const float* __restrict__ mean, const float* __restrict__ invstd,
                                                         double* __restrict__ partial, long long M, int C, long long rows_per_block,
                                                         const float* __restrict__ mscale = nullptr, const float* __restrict__ mbias = nullptr) {
    const int C4 = C >> 2;
    const int cols = C4 < 256 ? C4 : 256;                 // float4 columns handled by this block
    const int col = threadIdx.x % cols;
    const int rlane = threadIdx.x / cols, rstep = 256 / cols;
    const int c4 = blockIdx.y * 256 + col;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4 && rlane < rstep) {
        f32x4 mu = {0.f, 0.f, 0.f, 0.f}, is = {1.f, 1.f, 1.f, 1.f};
        f32x4 msc = {0.f, 0.f, 0.f, 0.f}, mbi = {1.f, 1.f, 1.f, 1.f};
        if (MODE == 1) { mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4); is = *reinterpret_cast<const f32x4*>(invstd + c4 * 4); }
        if (MODE == 1 && mscale) { msc = *reinterpret_cast<const f32x4*>(mscale + c4 * 4); mbi = *reinterpret_cast<const f32x4*>(mbias + c4 * 4); }
        const f32x4* __restrict__ a4 = reinterpret_cast<const f32x4*>(a);
        const f32x4* __restrict__ y4 = reinterpret_cast<const f32x4*>(y);
        const f32x4* __restrict__ z4 = reinterpret_cast<const f32x4*>(z);
        const long long C4l = C4;
        // rows are taken four at a time with all their loads issued first (a thread otherwise has one 16-byte load in
        // flight per step); the accumulation order is the plain row order either way
        auto fold = [&](f32x4 v, f32x4 yy, f32x4 zz) {
            if (MODE == 0) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[e] += v[e]; q[e] += v[e] * v[e]; }
            } else {
                if (y) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = yy[e] > 0.f ? v[e] : 0.f;
                }
                if (mscale) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = relu_on(zz[e], msc[e], mbi[e]) ? v[e] : 0.f;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[e] += v[e]; q[e] += v[e] * ((zz[e] - mu[e]) * is[e]); }
            }
        };
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        long long r = r0 + rlane;
        for (; r + 3LL * rstep < r1; r += 4LL * rstep) {
            f32x4 v[4], yy[4], zz[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long o = (r + (long long)u * rstep) * C4l + c4;
                v[u] = a4[o];
                yy[u] = (MODE == 1 && y) ? y4[o] : zero;
                zz[u] = MODE == 1 ? z4[o] : zero;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) fold(v[u], yy[u], zz[u]);
        }
        for (; r < r1; r += rstep) {
            const long long o = r * C4l + c4;
            fold(a4[o], (MODE == 1 && y) ? y4[o] : zero, MODE == 1 ? z4[o] : zero);
        }
    }
    __shared__ float sh[2][256][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[0][threadIdx.x][e] = s[e]; sh[1][threadIdx.x][e] = q[e]; }
    __syncthreads();
    if (rlane == 0 && c4 < C4) {
        double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
        for (int k = 0; k < rstep; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { ds[e] += sh[0][k * cols + col][e]; dq[e] += sh[1][k * cols + col][e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            partial[((long long)blockIdx.x * C + c4 * 4 + e) * 2 + 0] = ds[e];
            partial[((long long)blockIdx.x * C + c4 * 4 + e) * 2 + 1] = dq[e];
        }
    }
}

// scalar fallback for channel counts that are not a multiple of 4 (conv bias gradient of the 17-joint head is
// carried in 32 channels, so this is only a safety net)
__global__ __launch_bounds__(256) void col_stats_scalar_kernel(const float* __restrict__ z, double* __restrict__ partial, long long M, int C,
                                                               long long rows_per_block) {
    const int c = blockIdx.y * 64 + (threadIdx.x & 63);
    const int rl = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    float s = 0.f, q = 0.f;
    if (c < C)
        for (long long r = r0 + rl; r < r1; r += 4) { const float v = z[r * C + c]; s += v; q += v * v; }
    __shared__ double sh[2][4][64];
    sh[0][rl][threadIdx.x & 63] = s; sh[1][rl][threadIdx.x & 63] = q;
    __syncthreads();
    if (rl == 0 && c < C) {
        const int l = threadIdx.x & 63;
        partial[((long long)blockIdx.x * C + c) * 2 + 0] = sh[0][0][l] + sh[0][1][l] + sh[0][2][l] + sh[0][3][l];
        partial[((long long)blockIdx.x * C + c) * 2 + 1] = sh[1][0][l] + sh[1][1][l] + sh[1][2][l] + sh[1][3][l];
    }
}

// A 256-thread block owns four consecutive channels (c0 = 4 * blockIdx.x; wave w reports channel c0 + w): every thread walks
// the row-block partials rb = t, t + 256, ... and adds the four channels' (sum, sum^2) pairs — 64 contiguous bytes per row
// block — then the 256 per-thread sums are combined in a fixed order (xor-shuffle tree inside a wave, waves 0..3 in order).
// (The first version gave each channel ONE wave striding 16-byte pairs C*16 bytes apart: with C = 64 and 23 040 row blocks
// — the stem at B = 120 — sixteen blocks of four such waves took 159 us; this layout reads the same 23.6 MB in ~15 us.)
__device__ __forceinline__ void sum_partials(const double* __restrict__ partial, int nrb, int C, int c, int lane, double& s, double& q) {
    (void)c; (void)lane;
    const int c0 = blockIdx.x * 4;
    const int nc = C - c0 < 4 ? C - c0 : 4;
    double as[4] = {0.0, 0.0, 0.0, 0.0}, aq[4] = {0.0, 0.0, 0.0, 0.0};
    for (int rb = threadIdx.x; rb < nrb; rb += 256) {
        const double* __restrict__ src = partial + ((long long)rb * C + c0) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (e < nc) { as[e] += src[2 * e]; aq[e] += src[2 * e + 1]; }
    }
    __shared__ double sh[4][8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { as[e] = wave_sum(as[e]); aq[e] = wave_sum(aq[e]); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { sh[w][e] = as[e]; sh[w][4 + e] = aq[e]; }
    }
    __syncthreads();
    s = ((sh[0][w] + sh[1][w]) + sh[2][w]) + sh[3][w];
    q = ((sh[0][4 + w] + sh[1][4 + w]) + sh[2][4 + w]) + sh[3][4 + w];
}

__global__ __launch_bounds__(256) void bn_train_finalize_kernel(const double* __restrict__ partial, int nrb, long long M, int C,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float* __restrict__ running_mean, float* __restrict__ running_var, float momentum, float eps,
                                         float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                         float* __restrict__ scale, float* __restrict__ bias) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    double s, q;
    sum_partials(partial, nrb, C, c, threadIdx.x & 63, s, q);     // whole block takes part (barrier inside)
    if (c >= C || (threadIdx.x & 63) != 0) return;
    const double mean = s / (double)M;
    double var = q / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    save_mean[c] = (float)mean;
    save_invstd[c] = invstd;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    bias[c] = b - (float)mean * g * invstd;
    if (running_mean) {
        const double unbiased = M > 1 ? var * (double)M / (double)(M - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// y = act(z*scale[c] + bias[c] (+ res))
__global__ void scale_bias_act_kernel(const float* __restrict__ z, const float* __restrict__ scale, const float* __restrict__ bias,
                                      const float* __restrict__ res, float* __restrict__ y, long long n4, int C4, int relu) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const f32x4 v = *reinterpret_cast<const f32x4*>(z + i * 4);
        const f32x4 s = *reinterpret_cast<const f32x4*>(scale + c4 * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + c4 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaf(v[e], s[e], b[e]);
        if (res) {
            const f32x4 r = *reinterpret_cast<const f32x4*>(res + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] += r[e];
        }
        if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaxf(o[e], 0.f);
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = o;
    }
}

// dgamma, dbeta and the coefficients of dz = A*g + B*z + Cc
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int nrb, long long M, int C, const float* __restrict__ gamma,
                                       const float* __restrict__ mean, const float* __restrict__ invstd,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coefA,
                                       float* __restrict__ coefB, float* __restrict__ coefC) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    double sg, sgx;
    sum_partials(partial, nrb, C, c, threadIdx.x & 63, sg, sgx);
    if (c >= C || (threadIdx.x & 63) != 0) return;
    if (dbeta) dbeta[c] = (float)sg;
    if (dgamma) dgamma[c] = (float)sgx;
    const double s = (double)(gamma ? gamma[c] : 1.f) * (double)invstd[c];
    const double kb = -s * (double)invstd[c] * sgx / (double)M;
    coefA[c] = (float)s;
    coefB[c] = (float)kb;
    coefC[c] = (float)(-s * sg / (double)M - kb * (double)mean[c]);
}

// dz = A[c]*g + B[c]*z + C[c], g = dy*[y>0]; optionally also stores g (gradient of the skip connection)
__global__ void bn_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ z,
                                    const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Cc,
                                    float* __restrict__ dz, float* __restrict__ gout, long long n4, int C4,
                                    const float* __restrict__ mscale, const float* __restrict__ mbias) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + i * 4);
        if (y) {
            const f32x4 yy = *reinterpret_cast<const f32x4*>(y + i * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = yy[e] > 0.f ? g[e] : 0.f;
        }
        const f32x4 zz = *reinterpret_cast<const f32x4*>(z + i * 4);
        if (mscale) {
            const f32x4 msc = *reinterpret_cast<const f32x4*>(mscale + c4 * 4), mbi = *reinterpret_cast<const f32x4*>(mbias + c4 * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = relu_on(zz[e], msc[e], mbi[e]) ? g[e] : 0.f;
        }
        const f32x4 a = *reinterpret_cast<const f32x4*>(A + c4 * 4);
        const f32x4 b = *reinterpret_cast<const f32x4*>(B + c4 * 4);
        const f32x4 c = *reinterpret_cast<const f32x4*>(Cc + c4 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = a[e] * g[e] + b[e] * zz[e] + c[e];
        *reinterpret_cast<f32x4*>(dz + i * 4) = o;
        if (gout) *reinterpret_cast<f32x4*>(gout + i * 4) = g;
    }
}

// MaxPool2d(3,2,1) backward on NHWC: each input pixel collects dy of the windows whose FIRST maximum
// (row-major scan of the window, like ATen) it is.
__global__ void maxpool3x3s2_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                        int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * H * W * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const long long n = t / H;
        float acc = 0.f;
        // windows (oy, ox) with 2*oy-1 <= iy <= 2*oy+1
        for (int oy = (iy >> 1); oy <= ((iy + 1) >> 1); ++oy) {
            if (oy >= Ho) continue;
            for (int ox = (ix >> 1); ox <= ((ix + 1) >> 1); ++ox) {
                if (ox >= Wo) continue;
                float best = -INFINITY; int by = -1, bx = -1;
                for (int dy_ = 0; dy_ < 3; ++dy_) {
                    const int yy = 2 * oy - 1 + dy_;
                    if ((unsigned)yy >= (unsigned)H) continue;
                    for (int dx_ = 0; dx_ < 3; ++dx_) {
                        const int xx = 2 * ox - 1 + dx_;
                        if ((unsigned)xx >= (unsigned)W) continue;
                        const float v = x[((n * H + yy) * W + xx) * C + c];
                        if (v > best || by < 0) { best = v; by = yy; bx = xx; }
                    }
                }
                if (by == iy && bx == ix) acc += dy[((n * Ho + oy) * Wo + ox) * C + c];
            }
        }
        dx[i] = acc;
    }
}

// MaxPool2d(3,2,1) forward that also records which of the 9 window taps won (first maximum in scan order), and the
// backward that uses it: every input pixel looks at the <= 4 windows covering it (4 byte loads instead of 36 float loads)
__global__ void maxpool3x3s2_fwd_idx_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                            int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const long long n = t / Ho;
        float best = -INFINITY; int bk = 255;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int yy = 2 * oy - 1 + k / 3, xx = 2 * ox - 1 + k % 3;
            if ((unsigned)yy >= (unsigned)H || (unsigned)xx >= (unsigned)W) continue;
            const float v = x[((n * H + yy) * W + xx) * C + c];
            if (v > best || bk == 255) { best = v; bk = k; }
        }
        y[i] = best;
        idx[i] = (uint8_t)bk;
    }
}

__global__ void maxpool3x3s2_bwd_idx_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                            int N, int H, int W, int C, int Ho, int Wo) {
    const long long total = (long long)N * H * W * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const long long n = t / H;
        float acc = 0.f;
        for (int oy = (iy >> 1); oy <= ((iy + 1) >> 1); ++oy) {
            if (oy >= Ho) continue;
            for (int ox = (ix >> 1); ox <= ((ix + 1) >> 1); ++ox) {
                if (ox >= Wo) continue;
                const long long o = ((n * Ho + oy) * Wo + ox) * C + c;
                const int k = (iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1));      // this pixel's tap inside that window
                if (idx[o] == k) acc += dy[o];
            }
        }
        dx[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Stem tail of the ResNet trunks in training mode (Resnet.py:171-172: bn1 -> relu -> maxpool) without the full-resolution
// activation or its gradient ever being stored:
//   forward   pooled[o] = max over the 3x3/2 window of relu(z*scale + bias), idx[o] = winning tap (first maximum in scan
//             order): BatchNorm-affine and ReLU applied to z on load — the same fmaf as scale_bias_act_kernel, so values and
//             winners are those of the unfused pair of passes, bit for bit
//   backward  the gradient of a stem pixel is GATHERED from the <= 4 windows covering it (their idx byte says whether this
//             pixel won) wherever it is needed: once in the (sum g, sum g*xhat) reduction, once in dz = A*g + B*z + C.
// Four channels per thread (float4 / uchar4).
// ---------------------------------------------------------------------------------------------------------------------
template <bool AFFINE>
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_idx4_kernel(const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ bias,
                                                                   float* __restrict__ y, uint8_t* __restrict__ idx, int N, int H, int W, int C4, int Ho, int Wo) {
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const long long n = t / Ho;
        f32x4 sc = {1.f, 1.f, 1.f, 1.f}, bi = {0.f, 0.f, 0.f, 0.f};
        if (AFFINE) { sc = *reinterpret_cast<const f32x4*>(scale + c4 * 4); bi = *reinterpret_cast<const f32x4*>(bias + c4 * 4); }
        f32x4 v[9];
        bool ok[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) {                      // all nine loads in flight together
            const int yy = 2 * oy - 1 + k / 3, xx = 2 * ox - 1 + k % 3;
            ok[k] = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            v[k] = ok[k] ? *reinterpret_cast<const f32x4*>(x + (((n * H + yy) * W + xx) * C4 + c4) * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int bk[4] = {255, 255, 255, 255};
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (!ok[k]) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = AFFINE ? fmaxf(fmaf(v[k][e], sc[e], bi[e]), 0.f) : v[k][e];
                if (a > best[e] || bk[e] == 255) { best[e] = a; bk[e] = k; }
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = best;
        *reinterpret_cast<unsigned*>(idx + i * 4) = (unsigned)bk[0] | ((unsigned)bk[1] << 8) | ((unsigned)bk[2] << 16) | ((unsigned)bk[3] << 24);
    }
}

// gradient of stem pixel (n, iy, ix), channels 4*c4..: sum over the covering windows whose winner is this pixel
__device__ __forceinline__ f32x4 pool_gather4(const float* __restrict__ dy, const uint8_t* __restrict__ idx, long long n, int iy, int ix, int c4,
                                              int C4, int Ho, int Wo) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const int oy0 = iy >> 1, oy1 = (iy + 1) >> 1, ox0 = ix >> 1, ox1 = (ix + 1) >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int oy = a ? oy1 : oy0;
        if ((a && oy1 == oy0) || oy >= Ho) continue;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int ox = b ? ox1 : ox0;
            if ((b && ox1 == ox0) || ox >= Wo) continue;
            const long long o = ((n * Ho + oy) * Wo + ox) * C4 + c4;
            const unsigned kk = *reinterpret_cast<const unsigned*>(idx + o * 4);
            const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o * 4);
            const unsigned k = (unsigned)((iy - (2 * oy - 1)) * 3 + (ix - (2 * ox - 1)));      // this pixel's tap inside that window
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += ((kk >> (8 * e)) & 255u) == k ? d[e] : 0.f;
        }
    }
    return acc;
}

// max-pool backward alone (float4 version of maxpool3x3s2_bwd_idx_kernel; same sums, same order)
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_idx4_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                                                   int N, int H, int W, int C4, int Ho, int Wo) {
    const long long total = (long long)N * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        *reinterpret_cast<f32x4*>(dx + i * 4) = pool_gather4(dy, idx, t / H, iy, ix, c4, C4, Ho, Wo);
    }
}

// reduction pass: block = row range, thread = fixed float4 column (C4 <= 256 columns: the stem has 16), rows in order;
// partial[(rb*C + c)*2 + {0,1}] = (sum g, sum g*xhat), g = gathered gradient * [relu(z*msc + mbi) > 0]
__global__ __launch_bounds__(256) void pool_bn_bwd_reduce_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, const float* __restrict__ z,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ mscale, const float* __restrict__ mbias,
                                                                double* __restrict__ partial, long long M, int C, long long rows_per_block,
                                                                int H, int W, int Ho, int Wo) {
    const int C4 = C >> 2;
    const int cols = C4 < 256 ? C4 : 256;
    const int col = threadIdx.x % cols, rlane = threadIdx.x / cols, rstep = 256 / cols;
    const int c4 = blockIdx.y * 256 + col;
    const long long r0 = (long long)blockIdx.x * rows_per_block;
    const long long r1 = r0 + rows_per_block < M ? r0 + rows_per_block : M;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = {0.f, 0.f, 0.f, 0.f};
    if (c4 < C4 && rlane < rstep) {
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c4 * 4), is = *reinterpret_cast<const f32x4*>(invstd + c4 * 4);
        const f32x4 msc = *reinterpret_cast<const f32x4*>(mscale + c4 * 4), mbi = *reinterpret_cast<const f32x4*>(mbias + c4 * 4);
        for (long long r = r0 + rlane; r < r1; r += rstep) {
            const int ix = (int)(r % W);
            const long long t = r / W;
            const int iy = (int)(t % H);
            const f32x4 zz = *reinterpret_cast<const f32x4*>(z + (r * C4 + c4) * 4);
            const f32x4 g = pool_gather4(dy, idx, t / H, iy, ix, c4, C4, Ho, Wo);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = relu_on(zz[e], msc[e], mbi[e]) ? g[e] : 0.f;
                s[e] += v; q[e] += v * ((zz[e] - mu[e]) * is[e]);
            }
        }
    }
    __shared__ float sh[2][256][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { sh[0][threadIdx.x][e] = s[e]; sh[1][threadIdx.x][e] = q[e]; }
    __syncthreads();
    if (rlane == 0 && c4 < C4) {
        double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
        for (int k = 0; k < rstep; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { ds[e] += sh[0][k * cols + col][e]; dq[e] += sh[1][k * cols + col][e]; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            partial[((long long)blockIdx.x * C + c4 * 4 + e) * 2 + 0] = ds[e];
            partial[((long long)blockIdx.x * C + c4 * 4 + e) * 2 + 1] = dq[e];
        }
    }
}

__global__ __launch_bounds__(256) void pool_bn_bwd_apply_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, const float* __restrict__ z,
                                                               const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Cc,
                                                               const float* __restrict__ mscale, const float* __restrict__ mbias, float* __restrict__ dz,
                                                               long long n4, int C4, int H, int W, int Ho, int Wo) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int ix = (int)(t % W); t /= W;
        const int iy = (int)(t % H);
        const f32x4 zz = *reinterpret_cast<const f32x4*>(z + i * 4);
        const f32x4 g = pool_gather4(dy, idx, t / H, iy, ix, c4, C4, Ho, Wo);
        const f32x4 a = *reinterpret_cast<const f32x4*>(A + c4 * 4), b = *reinterpret_cast<const f32x4*>(B + c4 * 4), cc = *reinterpret_cast<const f32x4*>(Cc + c4 * 4);
        const f32x4 msc = *reinterpret_cast<const f32x4*>(mscale + c4 * 4), mbi = *reinterpret_cast<const f32x4*>(mbias + c4 * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = relu_on(zz[e], msc[e], mbi[e]) ? g[e] : 0.f;
            o[e] = a[e] * v + b[e] * zz[e] + cc[e];
        }
        *reinterpret_cast<f32x4*>(dz + i * 4) = o;
    }
}

// out[c] = sum over rows of x[r][c]  (conv bias gradient); same two-stage scheme
__global__ __launch_bounds__(256) void col_sum_finalize_kernel(const double* __restrict__ partial, int nrb, int C, float* __restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    double s, q;
    sum_partials(partial, nrb, C, c, threadIdx.x & 63, s, q);
    if (c < C && (threadIdx.x & 63) == 0) out[c] = (float)s;
}

static inline int ew_grid(long long n) { long long g = (n + 255) / 256; if (g > 4096) g = 4096; if (g < 1) g = 1; return (int)g; }

}  // namespace vatl

using namespace vatl;

extern "C" int64_t vatl_col_reduce_workspace_doubles(int64_t M, int C) { return 2 * (int64_t)row_split(M, C).nrb * (int64_t)C; }

static void launch_col_reduce(int mode, const float* a, const float* y, const float* z, const float* mean, const float* invstd,
                              double* ws, long long M, int C, const RowSplit& rs, hipStream_t st, const float* mscale = nullptr,
                              const float* mbias = nullptr) {
    if (C & 3) {                                              // stats only (mode 0) on odd channel counts
        hipLaunchKernelGGL(col_stats_scalar_kernel, dim3(rs.nrb, cdiv(C, 64)), dim3(256), 0, st, a, ws, M, C, rs.rows_per_block);
        return;
    }
    const dim3 grid(rs.nrb, cdiv(C / 4, 256));
    if (mode == 0) hipLaunchKernelGGL(col_reduce_kernel<0>, grid, dim3(256), 0, st, a, y, z, mean, invstd, ws, M, C, rs.rows_per_block,
                                      (const float*)nullptr, (const float*)nullptr);
    else           hipLaunchKernelGGL(col_reduce_kernel<1>, grid, dim3(256), 0, st, a, y, z, mean, invstd, ws, M, C, rs.rows_per_block, mscale, mbias);
}

extern "C" int vatl_bn_train_fwd_stats(const float* z, int64_t M, int C, const float* gamma, const float* beta, float* running_mean,
                                       float* running_var, float momentum, float eps, float* save_mean, float* save_invstd,
                                       float* scale, float* bias, double* workspace, void* stream) {
    if (!z || !save_mean || !save_invstd || !scale || !bias || !workspace || M <= 0) return fail(VATL_EINVAL, "bn_train_fwd_stats: bad arguments");
    const RowSplit rs = row_split(M, C);
    launch_col_reduce(0, z, nullptr, nullptr, nullptr, nullptr, workspace, (long long)M, C, rs, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_train_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, workspace, rs.nrb, (long long)M, C, gamma, beta,
                       running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, bias);
    return check_launch("bn_train_fwd_stats");
}

extern "C" int vatl_scale_bias_act(const float* z, const float* scale, const float* bias, const float* residual, float* y,
                                   int64_t M, int C, int relu, void* stream) {
    if (!z || !scale || !bias || !y || (C & 3)) return fail(VATL_EINVAL, "scale_bias_act: bad arguments");
    const long long n4 = M * C / 4;
    hipLaunchKernelGGL(scale_bias_act_kernel, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, z, scale, bias, residual, y, n4, C / 4, relu);
    return check_launch("scale_bias_act");
}

static int bn_train_bwd_impl(const float* dy, const float* y_or_null, const float* mscale, const float* mbias, const float* z, const float* gamma,
                             const float* save_mean, const float* save_invstd, float* dz, float* g_out_or_null, float* dgamma, float* dbeta,
                             int64_t M, int C, float* coef3C, double* workspace, void* stream) {
    if (!dy || !z || !save_mean || !save_invstd || !dz || !coef3C || !workspace || (C & 3) || M <= 0)
        return fail(VATL_EINVAL, "bn_train_bwd: bad arguments");
    const RowSplit rs = row_split(M, C);
    hipStream_t st = (hipStream_t)stream;
    launch_col_reduce(1, dy, y_or_null, z, save_mean, save_invstd, workspace, (long long)M, C, rs, st, mscale, mbias);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, workspace, rs.nrb, (long long)M, C, gamma, save_mean, save_invstd,
                       dgamma, dbeta, coef3C, coef3C + C, coef3C + 2 * C);
    const long long n4 = M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, st, dy, y_or_null, z, coef3C, coef3C + C, coef3C + 2 * C, dz, g_out_or_null, n4, C / 4,
                       mscale, mbias);
    return check_launch("bn_train_bwd");
}

extern "C" int vatl_bn_train_bwd(const float* dy, const float* y_or_null, const float* z, const float* gamma, const float* save_mean,
                                 const float* save_invstd, float* dz, float* g_out_or_null, float* dgamma, float* dbeta,
                                 int64_t M, int C, float* coef3C, double* workspace, void* stream) {
    return bn_train_bwd_impl(dy, y_or_null, nullptr, nullptr, z, gamma, save_mean, save_invstd, dz, g_out_or_null, dgamma, dbeta, M, C, coef3C, workspace, stream);
}

extern "C" int vatl_bn_train_bwd_relu(const float* dy, const float* scale, const float* bias, const float* z, const float* gamma,
                                      const float* save_mean, const float* save_invstd, float* dz, float* dgamma, float* dbeta,
                                      int64_t M, int C, float* coef3C, double* workspace, void* stream) {
    if (!scale || !bias) return fail(VATL_EINVAL, "bn_train_bwd_relu: null scale/bias");
    return bn_train_bwd_impl(dy, nullptr, scale, bias, z, gamma, save_mean, save_invstd, dz, nullptr, dgamma, dbeta, M, C, coef3C, workspace, stream);
}

// BatchNorm backward whose reduction pass already ran in the epilogue of the data-gradient launch that produced g
// (vatl_conv2d_fwd_ex_bnbwd): g = masked output gradient, partial = its (sum g, sum g*xhat) row-block partials.
// Finalize (dgamma, dbeta, coefficients) + ONE apply pass dz = A*g + B*z + C.
extern "C" int vatl_bn_bwd_from_stats(const double* partial, int64_t row_blocks, const float* g, const float* z, const float* gamma,
                                      const float* save_mean, const float* save_invstd, float* dz, float* dgamma, float* dbeta, int64_t M, int C,
                                      float* coef3C, void* stream) {
    if (!partial || !g || !z || !save_mean || !save_invstd || !dz || !coef3C || (C & 3) || M <= 0 || row_blocks <= 0 || row_blocks > 0x7FFFFFFF)
        return fail(VATL_EINVAL, "bn_bwd_from_stats: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, partial, (int)row_blocks, (long long)M, C, gamma, save_mean, save_invstd,
                       dgamma, dbeta, coef3C, coef3C + C, coef3C + 2 * C);
    const long long n4 = M * C / 4;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, st, g, (const float*)nullptr, z, coef3C, coef3C + C, coef3C + 2 * C, dz,
                       (float*)nullptr, n4, C / 4, (const float*)nullptr, (const float*)nullptr);
    return check_launch("bn_bwd_from_stats");
}

extern "C" int vatl_bn_train_finalize(const double* partial, int64_t row_blocks, int64_t M, int C, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                                      float* save_invstd, float* scale, float* bias, void* stream) {
    if (!partial || !save_mean || !save_invstd || !scale || !bias || M <= 0 || row_blocks <= 0 || row_blocks > 0x7FFFFFFF)
        return fail(VATL_EINVAL, "bn_train_finalize: bad arguments");
    hipLaunchKernelGGL(bn_train_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, partial, (int)row_blocks, (long long)M, C, gamma, beta,
                       running_mean, running_var, momentum, eps, save_mean, save_invstd, scale, bias);
    return check_launch("bn_train_finalize");
}

extern "C" int vatl_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
    if (!x || !dy || !dx) return fail(VATL_EINVAL, "maxpool3x3s2_bwd: null pointer");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(ew_grid((long long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, N, H, W, C, Ho, Wo);
    return check_launch("maxpool3x3s2_bwd");
}

extern "C" int vatl_col_sum(const float* x, int64_t M, int C, float* out, double* workspace, void* stream) {
    if (!x || !out || !workspace || M <= 0) return fail(VATL_EINVAL, "col_sum: bad arguments");
    const RowSplit rs = row_split(M, C);
    launch_col_reduce(0, x, nullptr, nullptr, nullptr, nullptr, workspace, (long long)M, C, rs, (hipStream_t)stream);
    hipLaunchKernelGGL(col_sum_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, (hipStream_t)stream, workspace, rs.nrb, C, out);
    return check_launch("col_sum");
}

static int maxpool_fwd_idx_impl(const float* x, const float* scale, const float* bias, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    if (!x || !y || !idx) return fail(VATL_EINVAL, "maxpool3x3s2_fwd_idx: null pointer");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if ((C & 3) == 0) {
        const dim3 grid(ew_grid((long long)N * Ho * Wo * (C / 4)));
        if (scale) hipLaunchKernelGGL(maxpool3x3s2_fwd_idx4_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, scale, bias, y, idx, N, H, W, C / 4, Ho, Wo);
        else       hipLaunchKernelGGL(maxpool3x3s2_fwd_idx4_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, scale, bias, y, idx, N, H, W, C / 4, Ho, Wo);
        return check_launch("maxpool3x3s2_fwd_idx");
    }
    if (scale) return fail(VATL_EINVAL, "maxpool3x3s2_fwd_idx_affine: C %d must be a multiple of 4", C);
    hipLaunchKernelGGL(maxpool3x3s2_fwd_idx_kernel, dim3(ew_grid((long long)N * Ho * Wo * C)), dim3(256), 0, (hipStream_t)stream, x, y, idx, N, H, W, C, Ho, Wo);
    return check_launch("maxpool3x3s2_fwd_idx");
}

extern "C" int vatl_maxpool3x3s2_fwd_idx(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream) {
    return maxpool_fwd_idx_impl(x, nullptr, nullptr, y, idx, N, H, W, C, stream);
}

// BatchNorm-affine + ReLU + MaxPool2d(3,2,1) in one pass over the conv output z (Resnet.py:171-172 in training mode):
// y = pool(relu(z*scale + bias)), idx = winning taps; relu(z*scale + bias) itself is never stored.
extern "C" int vatl_maxpool3x3s2_fwd_idx_affine(const float* z, const float* scale, const float* bias, float* y, uint8_t* idx, int N, int H, int W, int C,
                                                void* stream) {
    if (!scale || !bias) return fail(VATL_EINVAL, "maxpool3x3s2_fwd_idx_affine: null scale / bias");
    return maxpool_fwd_idx_impl(z, scale, bias, y, idx, N, H, W, C, stream);
}

// Backward of the same tail + the BatchNorm backward of the layer in front of it: dpool (N,Ho,Wo,C) = gradient of the pooled
// output, idx = the forward's winners; dz (N,H,W,C), dgamma, dbeta.  The full-resolution gradient is gathered on the fly in
// the reduction and in the apply pass (never stored).  workspace: vatl_col_reduce_workspace_doubles(N*H*W, C) doubles.
extern "C" int vatl_bn_train_bwd_relu_pool(const float* dpool, const uint8_t* idx, const float* scale, const float* bias, const float* z,
                                           const float* gamma, const float* save_mean, const float* save_invstd, float* dz, float* dgamma,
                                           float* dbeta, int N, int H, int W, int C, float* coef3C, double* workspace, void* stream) {
    if (!dpool || !idx || !scale || !bias || !z || !save_mean || !save_invstd || !dz || !coef3C || !workspace || (C & 3) || C > 1024 || N <= 0)
        return fail(VATL_EINVAL, "bn_train_bwd_relu_pool: bad arguments (C %% 4 == 0, C <= 1024)");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long long M = (long long)N * H * W;
    const RowSplit rs = row_split(M, C);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(pool_bn_bwd_reduce_kernel, dim3(rs.nrb, cdiv(C / 4, 256)), dim3(256), 0, st, dpool, idx, z, save_mean, save_invstd, scale, bias,
                       workspace, M, C, rs.rows_per_block, H, W, Ho, Wo);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 4)), dim3(256), 0, st, workspace, rs.nrb, M, C, gamma, save_mean, save_invstd, dgamma, dbeta,
                       coef3C, coef3C + C, coef3C + 2 * C);
    const long long n4 = M * C / 4;
    hipLaunchKernelGGL(pool_bn_bwd_apply_kernel, dim3(ew_grid(n4)), dim3(256), 0, st, dpool, idx, z, coef3C, coef3C + C, coef3C + 2 * C, scale, bias, dz, n4,
                       C / 4, H, W, Ho, Wo);
    return check_launch("bn_train_bwd_relu_pool");
}

extern "C" int vatl_maxpool3x3s2_bwd_idx(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C, void* stream) {
    if (!dy || !idx || !dx) return fail(VATL_EINVAL, "maxpool3x3s2_bwd_idx: null pointer");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    if ((C & 3) == 0) hipLaunchKernelGGL(maxpool3x3s2_bwd_idx4_kernel, dim3(ew_grid((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, N, H, W, C / 4, Ho, Wo);
    else hipLaunchKernelGGL(maxpool3x3s2_bwd_idx_kernel, dim3(ew_grid((long long)N * H * W * C)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, N, H, W, C, Ho, Wo);
    return check_launch("maxpool3x3s2_bwd_idx");
}
