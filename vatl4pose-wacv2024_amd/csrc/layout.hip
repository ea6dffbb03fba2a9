// Layout changes, parameter packing, pooling: the HBM-bound glue around the
// implicit-GEMM kernel.  All NHWC unless the name says otherwise.
#include "common.h"
#include "winograd_pack.h"

#include <cstring>

namespace vatl {

static thread_local char g_err[512] = "";
char* err_buf() { return g_err; }
int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static thread_local bool tl_meter_on = false;
static thread_local double tl_meter_flops[2] = {0.0, 0.0};
static thread_local long long tl_meter_launches[2] = {0, 0};
void meter_add(int kind, double flops) {
    if (!tl_meter_on) return;
    tl_meter_flops[kind & 1] += flops;
    tl_meter_launches[kind & 1] += 1;
}
static thread_local long long tl_meter_routes[kRouteCount] = {};
void meter_route(int route) {
    if (tl_meter_on && route >= 0 && route < kRouteCount) tl_meter_routes[route] += 1;
}
void meter_begin() {
    tl_meter_on = true; tl_meter_flops[0] = tl_meter_flops[1] = 0.0; tl_meter_launches[0] = tl_meter_launches[1] = 0;
    for (int k = 0; k < kRouteCount; ++k) tl_meter_routes[k] = 0;
}
int meter_routes(long long* out, int n) {
    for (int k = 0; k < n && k < kRouteCount; ++k) out[k] = tl_meter_routes[k];
    return kRouteCount;
}
void meter_end(double* flops, long long* launches) {
    for (int k = 0; k < 2; ++k) { flops[k] = tl_meter_flops[k]; launches[k] = tl_meter_launches[k]; }
    tl_meter_on = false;
}

// Cpad % 4 == 0, Cpad > 4 (the 17 -> 32 channel gradient of the heat-map head): one thread per (pixel, four channels), the channel
// group fastest, so a wave writes 1 KB of contiguous NHWC rows (the per-pixel version below wrote 4 bytes per lane 128 bytes apart:
// 165 us for 120 x 17 x 64 x 48 at 0.44 TB/s)
__global__ void nchw_to_nhwc_c4_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW, int Cpad) {
    const int G = Cpad >> 2;
    const long long total = (long long)N * HW * G;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int g = (int)(i % G);
        const long long pix = i / G;
        const long long n = pix / HW;
        const int px = (int)(pix - n * HW);
        const float* s = src + (n * C + 4 * g) * HW + px;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * g + k < C) v[k] = s[(long long)k * HW];
        *reinterpret_cast<f32x4*>(dst + i * 4) = v;
    }
}

// one thread per destination pixel; Cpad floats written contiguously
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW, int Cpad) {
    const long long total = (long long)N * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW;
        const int px = (int)(i - n * HW);
        const float* s = src + n * C * HW + px;
        float* d = dst + i * Cpad;
        if (Cpad == 4) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < C && c < 4; ++c) v[c] = s[(long long)c * HW];
            *reinterpret_cast<f32x4*>(d) = v;
        } else {
            for (int c = 0; c < Cpad; ++c) d[c] = c < C ? s[(long long)c * HW] : 0.f;
        }
    }
}

// one thread per destination element (n,c,pixel): reads stride C (L2 absorbs it; test/debug path)
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int C, int HW) {
    const long long total = (long long)N * C * HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int px = (int)(i % HW);
        const long long nc = i / HW;
        const int c = (int)(nc % C);
        const long long n = nc / C;
        dst[i] = src[(n * HW + px) * C + c];
    }
}

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int R, int S,
                                        int CoutPad, int Spad, int CinPad) {
    const long long total = (long long)CoutPad * R * Spad * CinPad;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % CinPad);
        long long t = i / CinPad;
        const int s = (int)(t % Spad); t /= Spad;
        const int r = (int)(t % R);
        const int o = (int)(t / R);
        float v = 0.f;
        if (o < Cout && s < S && c < Cin) v = w[(((long long)o * Cin + c) * R + r) * S + s];
        out[i] = v;
    }
}

// dual-source 1x1: out[o][k] = w1[o][k] * s1[o] (k < C1) | w2[o][k - C1] * s2[o]; rows o >= Cout zero; bias = b1 + b2
__global__ void pack_dual_weight_kernel(const float* __restrict__ w1, const float* __restrict__ s1, const float* __restrict__ b1,
                                        const float* __restrict__ w2, const float* __restrict__ s2, const float* __restrict__ b2,
                                        float* __restrict__ out, float* __restrict__ bias, int Cout, int C1, int C2, int CoutPad) {
    const int K = C1 + C2;
    const long long total = (long long)CoutPad * K;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K), o = (int)(i / K);
        float v = 0.f;
        if (o < Cout) v = k < C1 ? w1[(long long)o * C1 + k] * s1[o] : w2[(long long)o * C2 + (k - C1)] * s2[o];
        out[i] = v;
        if (k == 0 && o < Cout) bias[o] = b1[o] + b2[o];
    }
}

__global__ void pack_deconv_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cin, int Cout, int CoutPad) {
    const long long total = 4LL * CoutPad * 4 * Cin;     // [phase][CoutPad][ty][tx][Cin]
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cin);
        long long t = i / Cin;
        const int tx = (int)(t & 1); t >>= 1;
        const int ty = (int)(t & 1); t >>= 1;
        const int o = (int)(t % CoutPad);
        const int ph = (int)(t / CoutPad);
        const int ky = 3 - (ph >> 1) - 2 * ty, kx = 3 - (ph & 1) - 2 * tx;
        out[i] = o < Cout ? w[(((long long)c * Cout + o) * 4 + ky) * 4 + kx] : 0.f;
    }
}

__global__ void bn_fold_kernel(const float* gamma, const float* beta, const float* mean, const float* var, const float* cbias,
                               float eps, float* scale, float* bias, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 1.f, b = 0.f;
    if (var) {
        s = (gamma ? gamma[c] : 1.f) / sqrtf(var[c] + eps);
        b = (beta ? beta[c] : 0.f) - mean[c] * s;
    }
    if (cbias) b += cbias[c] * s;
    scale[c] = s;
    bias[c] = b;
}

// MaxPool2d(3,2,1): thread per (n, oy, ox, 4 channels); padding behaves as -inf
__global__ void maxpool3x3s2_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo) {
    const int C4 = C >> 2;
    const long long total = (long long)N * Ho * Wo * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const long long n = t / Ho;
        f32x4 m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int iy = oy * 2 - 1 + dy;
            if ((unsigned)iy >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int ix = ox * 2 - 1 + dx;
                if ((unsigned)ix >= (unsigned)W) continue;
                const f32x4 v = *reinterpret_cast<const f32x4*>(x + ((n * H + iy) * W + ix) * C + c4 * 4);
                m[0] = fmaxf(m[0], v[0]); m[1] = fmaxf(m[1], v[1]); m[2] = fmaxf(m[2], v[2]); m[3] = fmaxf(m[3], v[3]);
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = m;
    }
}

// global average pool: thread per (n, c), coalesced across c
__global__ void gap_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int HW, int C) {
    const long long total = (long long)N * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / C;
        const int c = (int)(i - n * C);
        const float* p = x + n * HW * C + c;
        float s = 0.f;
        for (int k = 0; k < HW; ++k) s += p[(long long)k * C];
        y[i] = s / (float)HW;
    }
}


// Per-item reductions over the pixels of an NHWC tensor for small batches (fine-tune steps: N = 32..120, where one thread per
// (item, channel) leaves most of the chip idle and walks thousands of pixels one load at a time).  A block owns `cols` float4
// channel columns of one item; its 256 threads take 256 / cols pixels at a time with four pixels in flight each and combine
// their partial sums through LDS in a fixed order (deterministic, no workspace, no atomics).
//   MODE 0: global average pool            out[n][c] = sum_hw x / HW
//   MODE 1: SE gate gradient               out[n][c] = sig'(gate) * sum_hw dy*[y>0]*u
template <int MODE>
__global__ __launch_bounds__(256) void hw_reduce_kernel(const float* __restrict__ a, const float* __restrict__ y, const float* __restrict__ u,
                                                        const float* __restrict__ gate, float* __restrict__ out, int HW, int C, int cols) {
    __shared__ f32x4 sh[256];
    const int C4 = C >> 2, groups = C4 / cols;
    const int n = blockIdx.x / groups, c4 = (blockIdx.x - n * groups) * cols + (threadIdx.x % cols);
    const int rlane = threadIdx.x / cols, rstep = 256 / cols;
    const long long base = (long long)n * HW * C4 + c4;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a) + base;
    const f32x4* y4 = reinterpret_cast<const f32x4*>(y) + base;
    const f32x4* u4 = reinterpret_cast<const f32x4*>(u) + base;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    auto fold = [&](f32x4 v, f32x4 yy, f32x4 uu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += MODE == 0 ? v[e] : (yy[e] > 0.f ? v[e] * uu[e] : 0.f);
    };
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    int k = rlane;
    for (; k + 3 * rstep < HW; k += 4 * rstep) {
        f32x4 v[4], yy[4], uu[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long o = (long long)(k + q * rstep) * C4;
            v[q] = a4[o];
            yy[q] = MODE == 1 ? y4[o] : zero;
            uu[q] = MODE == 1 ? u4[o] : zero;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) fold(v[q], yy[q], uu[q]);
    }
    for (; k < HW; k += rstep) {
        const long long o = (long long)k * C4;
        fold(a4[o], MODE == 1 ? y4[o] : zero, MODE == 1 ? u4[o] : zero);
    }
    sh[threadIdx.x] = s;
    __syncthreads();
    if (rlane == 0) {
        f32x4 t = sh[threadIdx.x];
        for (int r = 1; r < rstep; ++r) {
            const f32x4 w = sh[r * cols + threadIdx.x];
#pragma unroll
            for (int e = 0; e < 4; ++e) t[e] += w[e];
        }
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (MODE == 0) o[e] = t[e] / (float)HW;
            else { const float sg = 1.f / (1.f + expf(-gate[(long long)n * C + c4 * 4 + e])); o[e] = t[e] * sg * (1.f - sg); }
        }
        *reinterpret_cast<f32x4*>(out + (long long)n * C + c4 * 4) = o;
    }
}

// float4 columns per block for hw_reduce_kernel: the widest power of two (8 = 128-byte segments at least) that still gives >= 512
// blocks; 0 = use the thread-per-(item, channel) kernels
static int hw_reduce_cols(int N, int HW, int C) {
    if ((C & 31) || HW < 64) return 0;                          // few pixels per item: the thread-per-(item, channel) kernels are fine
    const int C4 = C >> 2;
    if (C4 & (C4 - 1)) return 0;
    int cols = 8;
    while (cols * 2 <= C4 && cols * 2 <= 256 && (long long)N * (C4 / (cols * 2)) >= 512) cols *= 2;
    return cols;
}

// PixelShuffle(2) on NHWC: out[b][2y+i][2x+j][c] = in[b][y][x][4c + 2i + j]
// thread per (input pixel, 4 input channels = one output channel c at the 4 sub-positions)
__global__ void pixelshuffle2_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long t = i / C4;
        const int px = (int)(t % W); t /= W;
        const int py = (int)(t % H);
        const long long n = t / H;
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        float* o = y + ((n * 2 * H + 2 * py) * 2 * W + 2 * px) * C4 + c;
        o[0] = v[0];
        o[C4] = v[1];
        o[(long long)2 * W * C4] = v[2];
        o[(long long)2 * W * C4 + C4] = v[3];
    }
}

// SE gate + residual + ReLU: y = relu(x * sigmoid(g[b][c]) + res)   (SE_module.py:20-24, SE_Resnet.py:125-135)
__global__ void se_scale_add_relu_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ res,
                                         float* __restrict__ y, int N, int HW, int C) {
    const int C4 = C >> 2;
    const long long total = (long long)N * HW * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long long n = i / ((long long)HW * C4);
        const f32x4 gv = *reinterpret_cast<const f32x4*>(g + n * C + c4 * 4);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(x + i * 4);
        const f32x4 rv = *reinterpret_cast<const f32x4*>(res + i * 4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = fmaxf(xv[e] * (1.f / (1.f + expf(-gv[e]))) + rv[e], 0.f);
        *reinterpret_cast<f32x4*>(y + i * 4) = o;
    }
}

// HRNet fuse: y = act(base + sum_k nearest_up(z_k, 2^shift_k)); up to 3 low-resolution sources (hrnet.py:242-260)
struct FuseUpArgs { const float* z[3]; int shift[3]; int n; };
__global__ void fuse_up_kernel(const float* __restrict__ base, FuseUpArgs a, float* __restrict__ y, int N, int H, int W, int C, int relu) {
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int px = (int)(t % W); t /= W;
        const int py = (int)(t % H);
        const long long n = t / H;
        f32x4 v = *reinterpret_cast<const f32x4*>(base + i * 4);
        for (int k = 0; k < a.n; ++k) {
            const int s = a.shift[k];
            const int h2 = H >> s, w2 = W >> s;
            const f32x4 z = *reinterpret_cast<const f32x4*>(a.z[k] + (((n * h2 + (py >> s)) * w2 + (px >> s)) * C4 + c4) * 4);
            v[0] += z[0]; v[1] += z[1]; v[2] += z[2]; v[3] += z[3];
        }
        if (relu) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        *reinterpret_cast<f32x4*>(y + i * 4) = v;
    }
}

// backward of the nearest up-sampling inside the HRNet fusion: dz[n][y][x][c] = sum over the 2^s x 2^s block of
// g = dy * [yact > 0] (yact = the fused, rectified output; NULL = no mask)
__global__ void upsample_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ yact, float* __restrict__ dz, int N, int H, int W,
                                    int C, int s) {
    const int C4 = C >> 2, h2 = H >> s, w2 = W >> s, f = 1 << s;
    const long long total = (long long)N * h2 * w2 * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long long t = i / C4;
        const int px = (int)(t % w2); t /= w2;
        const int py = (int)(t % h2);
        const long long n = t / h2;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < f; ++a)
            for (int b = 0; b < f; ++b) {
                const long long o = (((n * H + (py * f + a)) * W + (px * f + b)) * C4 + c4) * 4;
                f32x4 g = *reinterpret_cast<const f32x4*>(dy + o);
                if (yact) {
                    const f32x4 y = *reinterpret_cast<const f32x4*>(yact + o);
                    g[0] = y[0] > 0.f ? g[0] : 0.f; g[1] = y[1] > 0.f ? g[1] : 0.f; g[2] = y[2] > 0.f ? g[2] : 0.f; g[3] = y[3] > 0.f ? g[3] : 0.f;
                }
                acc[0] += g[0]; acc[1] += g[1]; acc[2] += g[2]; acc[3] += g[3];
            }
        *reinterpret_cast<f32x4*>(dz + i * 4) = acc;
    }
}

// backward of the global average pool: dx[n][p][c] = dy[n][c] / HW
__global__ void gap_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int HW, int C, float inv) {
    const int C4 = C >> 2;
    const long long total = (long long)N * HW * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long long n = i / ((long long)HW * C4);
        f32x4 g = *reinterpret_cast<const f32x4*>(dy + (n * C4 + c4) * 4);
        g[0] *= inv; g[1] *= inv; g[2] *= inv; g[3] *= inv;
        *reinterpret_cast<f32x4*>(dx + i * 4) = g;
    }
}

// data-gradient weights: out[c][t][n] = w[n][c][tap_r[t]][tap_s[t]]  ([CinPad][ntaps][Cout], rows c >= Cin zero)
struct TapList { int r[16]; int s[16]; int n; };
__global__ void pack_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int R, int S, int CinPad,
                                         int CoutK, TapList taps) {
    const long long total = (long long)CinPad * taps.n * CoutK;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int n = (int)(i % CoutK);
        long long t = i / CoutK;
        const int tp = (int)(t % taps.n);
        const int c = (int)(t / taps.n);
        out[i] = (c < Cin && n < Cout) ? w[(((long long)n * Cin + c) * R + taps.r[tp]) * S + taps.s[tp]] : 0.f;
    }
}


// Every weight re-pack of a fine-tune step in ONE launch (the trainers need ~60 .. 180 packed copies per step — forward
// layouts of the 3x3 / 7x7 / transposed convs, data-gradient layouts of every conv — and each used to be its own 5 us
// launch).  jobs: device array sorted by first_block; a block of 256 threads makes 1024 consecutive elements of one job (kinds 0 / 2),
// one 32 x 32 tile of one tap (kind 1: ceil(CinPad / 32) * ntaps * ceil(CoutK / 32) blocks) or 4096 elements of a Winograd filter
// transform (kinds 3 .. 6).
// Element arithmetic = pack_conv_weight_kernel / pack_dgrad_weight_kernel / pack_deconv_weight_kernel.
__global__ __launch_bounds__(256) void pack_multi_kernel(const VatlPackJob* __restrict__ jobs, int njobs) {
    __shared__ int sj;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;
        const long long b = blockIdx.x;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
        }
        sj = lo;
    }
    __syncthreads();
    const VatlPackJob* J = jobs + sj;
    const float* __restrict__ w = J->src;
    float* __restrict__ out = J->dst;
    const int kind = J->kind, Cout = J->Cout, Cin = J->Cin, R = J->R, S = J->S, pa = J->a, pb = J->b, pc = J->c;
    if (kind >= 7) {                                               // 7 / 8: F(4x4,3x3) filter transform, forward / data gradient; c = inner dimension of src; 256 items per block
        f4_pack_item(w, out, kind - 7, pc, Cout, Cin, ((long long)blockIdx.x - J->first_block) * 256 + threadIdx.x);
        return;
    }
    if (kind >= 3) {                                               // 3 / 4 / 5: Winograd filter transforms (forward, data gradient, transposed conv); b = NH, c = w_i
        wino_pack_block(w, out, kind - 3, pc, Cout, Cin, pb, (long long)blockIdx.x - J->first_block, threadIdx.x);   // 4096 elements per block
        return;
    }
    const unsigned bl = (unsigned)((long long)blockIdx.x - J->first_block);
    if (kind == 1) {
        // data-gradient layout [CinPad][ntaps][CoutK] = a transpose of OIHW: a block makes one 32 (input channels) x 32 (output channels)
        // tile of one tap through LDS, so that the reads run along the input channels of a filter row and the writes along the output
        // channels (one element per thread with 64-bit index arithmetic read a different cache line per lane: 305 us of the R50 step)
        __shared__ float tile[32][33];
        const unsigned tiles_n = (unsigned)(pb + 31) >> 5;
        const unsigned n_t = bl % tiles_n, t = bl / tiles_n;
        const unsigned tp = t % (unsigned)pc, c_t = t / (unsigned)pc;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        const int tap = J->tap_r[tp] * S + J->tap_s[tp], RS = R * S;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int n = (int)n_t * 32 + ty + 8 * k, c = (int)c_t * 32 + tx;
            tile[ty + 8 * k][tx] = (c < Cin && n < Cout) ? w[((size_t)n * Cin + c) * RS + tap] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = (int)c_t * 32 + ty + 8 * k, n = (int)n_t * 32 + tx;
            if (c < pa && n < pb) out[((size_t)c * pc + tp) * pb + n] = tile[tx][ty + 8 * k];
        }
        return;
    }
    // kinds 0 / 2 (forward layouts of the strided 3x3 / 7x7 convs and of the implicit-GEMM transposed convs): a few layers; 32-bit index arithmetic
    const unsigned base = bl * 1024u;
    const unsigned total = kind == 0 ? (unsigned)pa * R * pb * pc       // [CoutPad][R][Spad][CinPad]
                                     : 16u * pa * Cin;                  // [phase][CoutPad][ty][tx][Cin]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const unsigned i = base + e * 256 + threadIdx.x;
        if (i >= total) continue;
        float v = 0.f;
        if (kind == 0) {
            const unsigned c = i % (unsigned)pc;
            unsigned t = i / (unsigned)pc;
            const unsigned s_ = t % (unsigned)pb; t /= (unsigned)pb;
            const unsigned r = t % (unsigned)R;
            const unsigned o = t / (unsigned)R;
            if ((int)o < Cout && (int)s_ < S && (int)c < Cin) v = w[(((size_t)o * Cin + c) * R + r) * S + s_];
        } else {
            const unsigned c = i % (unsigned)Cin;
            unsigned t = i / (unsigned)Cin;
            const int tx = (int)(t & 1); t >>= 1;
            const int ty = (int)(t & 1); t >>= 1;
            const unsigned o = t % (unsigned)pa;
            const int ph = (int)(t / (unsigned)pa);
            const int ky = 3 - (ph >> 1) - 2 * ty, kx = 3 - (ph & 1) - 2 * tx;
            if ((int)o < Cout) v = w[(((size_t)c * Cout + o) * 4 + ky) * 4 + kx];
        }
        out[i] = v;
    }
}

// inverse of PixelShuffle(2) on NHWC (its backward): in (N,2H,2W,C/4) -> out (N,H,W,C), out[y][x][4c+2i+j] = in[2y+i][2x+j][c]
__global__ void pixelunshuffle2_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
    const int C4 = C >> 2;
    const long long total = (long long)N * H * W * C4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long t = i / C4;
        const int px = (int)(t % W); t /= W;
        const int py = (int)(t % H);
        const long long n = t / H;
        const float* s = x + ((n * 2 * H + 2 * py) * 2 * W + 2 * px) * C4 + c;
        f32x4 v;
        v[0] = s[0]; v[1] = s[C4]; v[2] = s[(long long)2 * W * C4]; v[3] = s[(long long)2 * W * C4 + C4];
        *reinterpret_cast<f32x4*>(y + i * 4) = v;
    }
}

// SE block backward, stage 1 (per item and channel): gm = dy*[y>0];  dgate[n][c] = sig'(g) * sum_hw gm*u
__global__ void se_bwd_gate_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ u,
                                   const float* __restrict__ gate, float* __restrict__ dgate, int N, int HW, int C) {
    const long long total = (long long)N * C;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / C;
        const int c = (int)(i - n * C);
        const long long base = n * HW * C + c;
        float s = 0.f;
        for (int k = 0; k < HW; ++k) {
            const long long o = base + (long long)k * C;
            if (y[o] > 0.f) s += dy[o] * u[o];
        }
        const float sg = 1.f / (1.f + expf(-gate[i]));
        dgate[i] = s * sg * (1.f - sg);
    }
}

// SE block backward, stage 2: gm = dy*[y>0] (gradient of the shortcut), du = gm*sigmoid(g) + dpool[n][c]/HW
__global__ void se_bwd_apply_kernel(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ gate,
                                    const float* __restrict__ dpool, float* __restrict__ du, float* __restrict__ gm, int N, int HW, int C) {
    const int C4 = C >> 2;
    const long long total = (long long)N * HW * C4;
    const float inv = 1.f / (float)HW;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long long n = i / ((long long)HW * C4);
        const f32x4 g = *reinterpret_cast<const f32x4*>(gate + n * C + c4 * 4);
        const f32x4 dp = *reinterpret_cast<const f32x4*>(dpool + n * C + c4 * 4);
        const f32x4 d = *reinterpret_cast<const f32x4*>(dy + i * 4);
        const f32x4 yy = *reinterpret_cast<const f32x4*>(y + i * 4);
        f32x4 o, m;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            m[e] = yy[e] > 0.f ? d[e] : 0.f;
            o[e] = m[e] * (1.f / (1.f + expf(-g[e]))) + dp[e] * inv;
        }
        *reinterpret_cast<f32x4*>(du + i * 4) = o;
        *reinterpret_cast<f32x4*>(gm + i * 4) = m;
    }
}

// dx = dy * [y > 0]   (ReLU backward on a flat fp32 span)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

static inline int grid_for(long long total, int block = 256) {
    long long g = (total + block - 1) / block;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace vatl

using namespace vatl;

extern "C" int vatl_version(void) { return VATL_VERSION; }
extern "C" const char* vatl_last_error(void) { return err_buf(); }

namespace vatl { void meter_begin(); void meter_end(double*, long long*); int meter_routes(long long*, int); }
extern "C" int vatl_flop_meter_begin(void) { vatl::meter_begin(); return 0; }
extern "C" int vatl_flop_meter_end(double* direct_flops, double* winograd_flops, int64_t* direct_launches, int64_t* winograd_launches) {
    double f[2]; long long n[2];
    vatl::meter_end(f, n);
    if (direct_flops) *direct_flops = f[0];
    if (winograd_flops) *winograd_flops = f[1];
    if (direct_launches) *direct_launches = n[0];
    if (winograd_launches) *winograd_launches = n[1];
    return 0;
}
extern "C" int vatl_flop_meter_routes(int64_t* counts, int n) {
    if (!counts || n < 0) return vatl::fail(VATL_EINVAL, "flop_meter_routes: null pointer");
    long long tmp[64] = {};
    const int have = vatl::meter_routes(tmp, n < 64 ? n : 64);
    for (int k = 0; k < n; ++k) counts[k] = k < have && k < 64 ? (int64_t)tmp[k] : 0;
    return have;
}

extern "C" int vatl_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int Cpad, void* stream) {
    if (!src || !dst || Cpad < C) return fail(VATL_EINVAL, "nchw_to_nhwc: bad arguments");
    if (Cpad > 4 && (Cpad & 3) == 0 && ((uintptr_t)dst & 15) == 0)
        hipLaunchKernelGGL(nchw_to_nhwc_c4_kernel, dim3(grid_for((long long)N * H * W * (Cpad / 4))), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, H * W, Cpad);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for((long long)N * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, H * W, Cpad);
    return check_launch("nchw_to_nhwc");
}

extern "C" int vatl_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, void* stream) {
    if (!src || !dst) return fail(VATL_EINVAL, "nhwc_to_nchw: null pointer");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for((long long)N * C * H * W)), dim3(256), 0, (hipStream_t)stream, src, dst, N, C, H * W);
    return check_launch("nhwc_to_nchw");
}

extern "C" int vatl_pack_conv_weight(const float* w, float* out, int Cout, int Cin, int R, int S, int CoutPad, int Spad, int CinPad, void* stream) {
    if (!w || !out || CoutPad < Cout || Spad < S || CinPad < Cin) return fail(VATL_EINVAL, "pack_conv_weight: bad arguments");
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3(grid_for((long long)CoutPad * R * Spad * CinPad)), dim3(256), 0, (hipStream_t)stream,
                       w, out, Cout, Cin, R, S, CoutPad, Spad, CinPad);
    return check_launch("pack_conv_weight");
}

extern "C" int vatl_pack_conv1x1_dual_weight(const float* w1, const float* scale1, const float* bias1, const float* w2, const float* scale2,
                                             const float* bias2, float* out, float* bias, int Cout, int C1, int C2, int CoutPad, void* stream) {
    if (!w1 || !scale1 || !bias1 || !w2 || !scale2 || !bias2 || !out || !bias || CoutPad < Cout) return fail(VATL_EINVAL, "pack_conv1x1_dual_weight: bad arguments");
    hipLaunchKernelGGL(pack_dual_weight_kernel, dim3(grid_for((long long)CoutPad * (C1 + C2))), dim3(256), 0, (hipStream_t)stream, w1, scale1, bias1, w2,
                       scale2, bias2, out, bias, Cout, C1, C2, CoutPad);
    return check_launch("pack_conv1x1_dual_weight");
}

extern "C" int vatl_pack_deconv4x4s2_weight(const float* w, float* out, int Cin, int Cout, int CoutPad, void* stream) {
    if (!w || !out || CoutPad < Cout) return fail(VATL_EINVAL, "pack_deconv4x4s2_weight: bad arguments");
    hipLaunchKernelGGL(pack_deconv_weight_kernel, dim3(grid_for(16LL * CoutPad * Cin)), dim3(256), 0, (hipStream_t)stream, w, out, Cin, Cout, CoutPad);
    return check_launch("pack_deconv4x4s2_weight");
}

extern "C" int vatl_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, const float* conv_bias,
                            float eps, float* scale, float* bias, int C, void* stream) {
    if (!scale || !bias || (var && !mean)) return fail(VATL_EINVAL, "bn_fold: bad arguments");
    hipLaunchKernelGGL(bn_fold_kernel, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, mean, var, conv_bias, eps, scale, bias, C);
    return check_launch("bn_fold");
}

extern "C" int vatl_maxpool3x3s2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if (!x || !y || (C & 3)) return fail(VATL_EINVAL, "maxpool3x3s2_fwd: C %d must be a multiple of 4", C);
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for((long long)N * Ho * Wo * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C, Ho, Wo);
    return check_launch("maxpool3x3s2_fwd");
}

extern "C" int vatl_gap_fwd(const float* x, float* y, int N, int HW, int C, void* stream) {
    if (!x || !y) return fail(VATL_EINVAL, "gap_fwd: null pointer");
    if (const int cols = hw_reduce_cols(N, HW, C)) {
        hipLaunchKernelGGL(hw_reduce_kernel<0>, dim3((unsigned)(N * ((C >> 2) / cols))), dim3(256), 0, (hipStream_t)stream, x, x, x, (const float*)nullptr, y, HW, C, cols);
        return check_launch("gap_fwd");
    }
    hipLaunchKernelGGL(gap_kernel, dim3(grid_for((long long)N * C)), dim3(256), 0, (hipStream_t)stream, x, y, N, HW, C);
    return check_launch("gap_fwd");
}

extern "C" int vatl_pixelshuffle2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if (!x || !y || (C & 15)) return fail(VATL_EINVAL, "pixelshuffle2_fwd: C %d must be a multiple of 16", C);
    hipLaunchKernelGGL(pixelshuffle2_kernel, dim3(grid_for((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C);
    return check_launch("pixelshuffle2_fwd");
}

extern "C" int vatl_se_scale_add_relu(const float* x, const float* gate, const float* residual, float* y, int N, int HW, int C, void* stream) {
    if (!x || !gate || !residual || !y || (C & 3)) return fail(VATL_EINVAL, "se_scale_add_relu: bad arguments");
    hipLaunchKernelGGL(se_scale_add_relu_kernel, dim3(grid_for((long long)N * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, gate, residual, y, N, HW, C);
    return check_launch("se_scale_add_relu");
}

extern "C" int vatl_fuse_upsample_add(const float* base, const float* z0, int shift0, const float* z1, int shift1, const float* z2, int shift2,
                                      float* y, int N, int H, int W, int C, int relu, void* stream) {
    if (!base || !y || (C & 3)) return fail(VATL_EINVAL, "fuse_upsample_add: bad arguments");
    FuseUpArgs a{};
    const float* zs[3] = {z0, z1, z2};
    const int sh[3] = {shift0, shift1, shift2};
    for (int k = 0; k < 3; ++k) {
        if (!zs[k]) continue;
        if (sh[k] < 1 || (H & ((1 << sh[k]) - 1)) || (W & ((1 << sh[k]) - 1)))
            return fail(VATL_EINVAL, "fuse_upsample_add: %dx%d is not divisible by 2^%d", H, W, sh[k]);
        a.z[a.n] = zs[k]; a.shift[a.n] = sh[k]; ++a.n;
    }
    hipLaunchKernelGGL(fuse_up_kernel, dim3(grid_for((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, base, a, y, N, H, W, C, relu);
    return check_launch("fuse_upsample_add");
}

extern "C" int vatl_upsample_nearest_bwd(const float* dy, const float* yact_or_null, float* dz, int N, int H, int W, int C, int shift, void* stream) {
    if (!dy || !dz || (C & 3)) return fail(VATL_EINVAL, "upsample_nearest_bwd: bad arguments");
    if (shift < 1 || shift > 5 || (H & ((1 << shift) - 1)) || (W & ((1 << shift) - 1)))
        return fail(VATL_EINVAL, "upsample_nearest_bwd: %dx%d is not divisible by 2^%d", H, W, shift);
    if (N <= 0) return 0;
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3(grid_for((long long)N * (H >> shift) * (W >> shift) * (C / 4))), dim3(256), 0, (hipStream_t)stream,
                       dy, yact_or_null, dz, N, H, W, C, shift);
    return check_launch("upsample_nearest_bwd");
}

extern "C" int vatl_gap_bwd(const float* dy, float* dx, int N, int HW, int C, void* stream) {
    if (!dy || !dx || (C & 3) || HW < 1) return fail(VATL_EINVAL, "gap_bwd: bad arguments");
    if (N <= 0) return 0;
    hipLaunchKernelGGL(gap_bwd_kernel, dim3(grid_for((long long)N * HW * (C / 4))), dim3(256), 0, (hipStream_t)stream, dy, dx, N, HW, C, 1.0f / (float)HW);
    return check_launch("gap_bwd");
}

extern "C" int vatl_pack_dgrad_weight(const float* w_oihw, float* out, int Cout, int Cin, int R, int S, int CinPad, int CoutK,
                                      int ntaps, const int* tap_r, const int* tap_s, void* stream) {
    if (!w_oihw || !out || !tap_r || !tap_s || ntaps < 1 || ntaps > 16 || CinPad < Cin || CoutK < Cout) return fail(VATL_EINVAL, "pack_dgrad_weight: bad arguments");
    TapList t{};
    t.n = ntaps;
    for (int i = 0; i < ntaps; ++i) {
        if (tap_r[i] < 0 || tap_r[i] >= R || tap_s[i] < 0 || tap_s[i] >= S) return fail(VATL_EINVAL, "pack_dgrad_weight: tap %d out of range", i);
        t.r[i] = tap_r[i]; t.s[i] = tap_s[i];
    }
    hipLaunchKernelGGL(pack_dgrad_weight_kernel, dim3(grid_for((long long)CinPad * ntaps * CoutK)), dim3(256), 0, (hipStream_t)stream, w_oihw, out, Cout, Cin, R, S, CinPad, CoutK, t);
    return check_launch("pack_dgrad_weight");
}

extern "C" int vatl_pack_weights_multi(const VatlPackJob* jobs_device, int njobs, int64_t total_blocks, void* stream) {
    if (njobs == 0) return 0;
    if (!jobs_device || njobs < 0 || total_blocks <= 0 || total_blocks > 0x7FFFFFFF) return fail(VATL_EINVAL, "pack_weights_multi: bad arguments");
    hipLaunchKernelGGL(pack_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device, njobs);
    return check_launch("pack_weights_multi");
}

extern "C" int vatl_pixelunshuffle2(const float* x, float* y, int N, int H, int W, int C, void* stream) {
    if (!x || !y || (C & 15)) return fail(VATL_EINVAL, "pixelunshuffle2: C %d must be a multiple of 16", C);
    hipLaunchKernelGGL(pixelunshuffle2_kernel, dim3(grid_for((long long)N * H * W * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C);
    return check_launch("pixelunshuffle2");
}

extern "C" int vatl_se_bwd(const float* dy, const float* y, const float* u, const float* gate, const float* dpool_or_null,
                           float* dgate_or_null, float* du_or_null, float* gm_or_null, int N, int HW, int C, void* stream) {
    if (!dy || !y || !gate || (C & 3)) return fail(VATL_EINVAL, "se_bwd: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (dgate_or_null) {
        if (!u) return fail(VATL_EINVAL, "se_bwd: stage 1 needs u");
        if (const int cols = hw_reduce_cols(N, HW, C))
            hipLaunchKernelGGL(hw_reduce_kernel<1>, dim3((unsigned)(N * ((C >> 2) / cols))), dim3(256), 0, st, dy, y, u, gate, dgate_or_null, HW, C, cols);
        else
        hipLaunchKernelGGL(se_bwd_gate_kernel, dim3(grid_for((long long)N * C)), dim3(256), 0, st, dy, y, u, gate, dgate_or_null, N, HW, C);
    }
    if (du_or_null) {
        if (!dpool_or_null || !gm_or_null) return fail(VATL_EINVAL, "se_bwd: stage 2 needs dpool and gm");
        hipLaunchKernelGGL(se_bwd_apply_kernel, dim3(grid_for((long long)N * HW * (C / 4))), dim3(256), 0, st, dy, y, gate, dpool_or_null, du_or_null, gm_or_null, N, HW, C);
    }
    return check_launch("se_bwd");
}

extern "C" int vatl_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream) {
    if (!dy || !y || !dx) return fail(VATL_EINVAL, "relu_bwd: null pointer");
    if (n <= 0) return 0;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, dy, y, dx, (long long)n);
    return check_launch("relu_bwd");
}
