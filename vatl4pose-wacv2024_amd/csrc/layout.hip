// Layout changes, parameter packing, pooling: the HBM-bound glue around the
// implicit-GEMM kernel.  All NHWC unless the name says otherwise.
#include "common.h"

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

extern "C" int vatl_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int Cpad, void* stream) {
    if (!src || !dst || Cpad < C) return fail(VATL_EINVAL, "nchw_to_nhwc: bad arguments");
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
    hipLaunchKernelGGL(gap_kernel, dim3(grid_for((long long)N * C)), dim3(256), 0, (hipStream_t)stream, x, y, N, HW, C);
    return check_launch("gap_fwd");
}
