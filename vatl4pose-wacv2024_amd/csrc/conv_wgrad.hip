// Weight gradient of Conv2d / ConvTranspose2d(4,2,1) as an implicit GEMM on the fp32 matrix cores.
//
//   dW[n][t][c] = sum_m G[m][n] * X[pix(m) + tap t][c]
//   G  (M, Cn)      "plain" operand, NHWC rows m = (b, oy, ox): the gradient of the conv output
//                   (for the transposed conv: the layer INPUT, Cn = Cin of the deconv)
//   X  (N,H,W,Cx)   "gathered" operand: the conv input (for the transposed conv: the output gradient)
//   t = (r, s)      filter tap, pix(m)+t = (oy*stride - pad + r, ox*stride - pad + s), zero outside
//
// GEMM shape: rows n (Cn), cols j = (t, c) (R*S*Cx), reduction over the M pixels — the long axis —
// so the launch is split along M and partial tiles are combined with fp32 atomics into the packed
// gradient [Cn][R*S][Cx] (zeroed first; summation order across splits is not fixed: ~1e-7 relative
// run-to-run noise, documented in DESIGN.md).  Both operands are staged K-major ([m][n] / [m][c], the
// natural NHWC layout), so one ds_read_b32 per operand tile feeds an MFMA k-step (lanes 0-31 take
// pixel 2s, lanes 32-63 pixel 2s+1).  v_mfma_f32_32x32x2_f32, exact fp32.
#include "common.h"

#include <atomic>

namespace vatl {

struct WgradParams {
    const float* g;      // (M, Cn)
    const float* x;      // (N, H, W, Cx)
    float* dw;           // [Cn][R*S (stem: R*8)][Cx] packed, fp32, pre-zeroed
    int M, Cn, Cx;       // Cn = valid rows (output channels), Cx = channels of the gathered operand
    int Gs;              // row stride of g in floats (>= Cn; extra channels are padding)
    int N, H, W;         // gathered image
    int Ho, Wo;          // pixel grid of the plain operand (M = N*Ho*Wo)
    int R, S, stride, pad;
    int ktiles;          // ceil(M / 32)
    int kt_per_split;
    int n_tiles, j_tiles_per_tap, taps;
    unsigned g_bytes, x_bytes;
};

constexpr unsigned WG_OOB = 0xFFFFFFFFu;
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 wg_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// BN: rows (n) per block, BJ: cols (channels of one tap) per block; wave tile WN x WJ; STEM: Cx = 4, one
// block column = one filter row (8 taps x 4 channels)
template <int BN, int BJ, int WN, int WJ, bool STEM>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradParams p) {
    constexpr int NWN = BN / WN, NWJ = BJ / WJ, NWK = 4 / (NWN * NWJ);
    static_assert(NWN * NWJ * NWK == 4 && NWK >= 1, "4 waves");
    constexpr int TN = WN / 32, TJ = WJ / 32;
    constexpr int LG = BN / 32, LX = BJ / 32;            // float4 loads per thread per k-tile (32 rows each)
    __shared__ __attribute__((aligned(16))) float Gs[2][32][BN];
    __shared__ __attribute__((aligned(16))) float Xs[2][32][BJ];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave / (NWN * NWJ), wr = wave % (NWN * NWJ);
    const int wn = wr / NWJ, wj = wr % NWJ;

    // block -> (n tile, tap, channel tile, M split)
    int bid = blockIdx.x;
    const int n_tile = bid % p.n_tiles; bid /= p.n_tiles;
    const int jt = bid % p.j_tiles_per_tap; bid /= p.j_tiles_per_tap;
    const int tap = bid % p.taps;
    const int split = bid / p.taps;
    const int n0 = n_tile * BN;
    const int c0 = jt * BJ;
    const int r = STEM ? tap : tap / p.S, s = STEM ? 0 : tap - r * p.S;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(kt0 + p.kt_per_split, p.ktiles);
    if (kt0 >= kt1) return;

    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, p.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const int HoWo = p.Ho * p.Wo;

    f32x4 rg[LG], rx[LX];
    auto gload = [&](int kt) {
        const int mb = kt * 32;
#pragma unroll
        for (int q = 0; q < LG; ++q) {
            const int f = tid + 256 * q;
            const int row = f / (BN / 4), c4 = f % (BN / 4);
            const int m = mb + row, n = n0 + c4 * 4;
            rg[q] = wg_load4(gr, (m < p.M && n < p.Gs) ? (unsigned)(m * p.Gs + n) << 2 : WG_OOB);
        }
#pragma unroll
        for (int q = 0; q < LX; ++q) {
            const int f = tid + 256 * q;
            const int row = f / (BJ / 4), c4 = f % (BJ / 4);
            const int m = mb + row;
            const int b = m / HoWo;
            const int rem = m - b * HoWo;
            const int oy = rem / p.Wo;
            const int ox = rem - oy * p.Wo;
            const int iy = oy * p.stride - p.pad + r;
            int ix = ox * p.stride - p.pad + s;
            int c = c0 + c4 * 4;
            if (STEM) { ix += c4; c = 0; }                 // float4 = one of the 8 taps of this filter row, 4 channels
            const bool ok = m < p.M && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W && c < p.Cx;
            rx[q] = wg_load4(xr, ok ? (unsigned)(((b * p.H + iy) * p.W + ix) * p.Cx + c) << 2 : WG_OOB);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < LG; ++q) {
            const int f = tid + 256 * q;
            *reinterpret_cast<f32x4*>(&Gs[buf][f / (BN / 4)][(f % (BN / 4)) * 4]) = rg[q];
        }
#pragma unroll
        for (int q = 0; q < LX; ++q) {
            const int f = tid + 256 * q;
            *reinterpret_cast<f32x4*>(&Xs[buf][f / (BJ / 4)][(f % (BJ / 4)) * 4]) = rx[q];
        }
    };

    f32x16 acc[TN][TJ];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    gload(kt0);
    lstore(0);
    __syncthreads();
    const int half = lane >> 5, l31 = lane & 31;
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        const bool more = kt + 1 < kt1;
        gload(more ? kt + 1 : kt);                         // tail: harmless reload, keeps the body branch-free
#pragma unroll
        for (int st = 0; st < 16 / NWK; ++st) {
            const int row = 2 * (st * NWK + wk) + half;    // pixel of this k-step handled by this lane half
            float a[TN], b[TJ];
#pragma unroll
            for (int i = 0; i < TN; ++i) a[i] = Gs[buf][row][wn * WN + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = Xs[buf][row][wj * WJ + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        lstore(buf ^ 1);
        __syncthreads();
    }

    // D[row = (e&3) + 8*(e>>2) + 4*(lane>>5)] -> n,  [col = lane&31] -> channel: coalesced fp32 atomics
    const int Kp = (STEM ? p.R * 8 : p.R * p.S) * p.Cx;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int cj = wj * WJ + j * 32 + l31;                    // column inside the block tile
            const int col = STEM ? tap * 32 + cj : tap * p.Cx + c0 + cj;
            const bool cv = STEM ? true : (c0 + cj) < p.Cx;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * WN + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (cv && n < p.Cn) atomicAdd(p.dw + (long long)n * Kp + col, acc[i][j][e]);
            }
        }
}

template <int BN, int BJ, int WN, int WJ, bool STEM>
static int launch_wgrad(WgradParams p, hipStream_t st) {
    p.n_tiles = cdiv(p.Cn, BN);
    p.taps = STEM ? p.R : p.R * p.S;
    p.j_tiles_per_tap = STEM ? 1 : cdiv(p.Cx, BJ);
    const int tiles = p.n_tiles * p.taps * p.j_tiles_per_tap;
    // enough M-splits for ~2048 blocks, at least 4 k-tiles each
    int splits = (2048 + tiles - 1) / tiles;
    const int max_splits = (p.ktiles + 3) / 4;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    p.kt_per_split = (p.ktiles + splits - 1) / splits;
    splits = (p.ktiles + p.kt_per_split - 1) / p.kt_per_split;
    hipLaunchKernelGGL((conv_wgrad_kernel<BN, BJ, WN, WJ, STEM>), dim3((unsigned)(tiles * splits)), dim3(256), 0, st, p);
    return check_launch("conv_wgrad");
}

// out (Cout,Cin,R,S) <- packed [Cout][R][Spad][CinPad]   (inverse of pack_conv_weight, drops the padding)
__global__ void unpack_conv_grad_kernel(const float* __restrict__ packed, float* __restrict__ out, int Cout, int Cin, int R, int S, int Spad, int CinPad) {
    const long long total = (long long)Cout * Cin * R * S;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(i % S);
        long long t = i / S;
        const int r = (int)(t % R); t /= R;
        const int c = (int)(t % Cin);
        const int o = (int)(t / Cin);
        out[i] = packed[(((long long)o * R + r) * Spad + s) * CinPad + c];
    }
}

// out (Cin,Cout,4,4) <- packed [Cin][ky][kx][Cout]
__global__ void unpack_deconv_grad_kernel(const float* __restrict__ packed, float* __restrict__ out, int Cin, int Cout) {
    const long long total = (long long)Cin * Cout * 16;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % 16);
        long long t = i / 16;
        const int o = (int)(t % Cout);
        const int c = (int)(t / Cout);
        out[i] = packed[((long long)c * 16 + k) * Cout + o];
    }
}

}  // namespace vatl

using namespace vatl;

extern "C" int64_t vatl_conv2d_wgrad_workspace_floats(int Cout, int Cin, int R, int S) {
    return Cin == 3 ? (int64_t)Cout * R * 8 * 4 : (int64_t)Cout * R * S * Cin;
}

// dw (Cout,Cin,R,S) = sum over pixels of dz (x) gathered input.  x NHWC (N,H,W,Cin) (Cin = 4 padded for the
// 3-channel stem), dz NHWC (N,Ho,Wo,CoutG) where CoutG >= Cout is the channel stride of dz.
extern "C" int vatl_conv2d_wgrad(const float* x, const float* dz, float* dw, float* workspace, int N, int H, int W, int Cin,
                                 int Cout, int CoutG, int R, int S, int stride, int pad, void* stream) {
    if (!x || !dz || !dw || !workspace) return fail(VATL_EINVAL, "conv2d_wgrad: null pointer");
    const bool stem = (Cin == 3);
    const int Cx = stem ? 4 : Cin;
    if (!stem && (Cin % 4)) return fail(VATL_EINVAL, "conv2d_wgrad: Cin %d must be a multiple of 4 (or 3 for the stem)", Cin);
    if ((CoutG % 4) || CoutG < Cout) return fail(VATL_EINVAL, "conv2d_wgrad: gradient channel stride %d invalid for Cout %d", CoutG, Cout);
    if (stem && S > 8) return fail(VATL_EINVAL, "conv2d_wgrad: stem filter width %d > 8", S);
    WgradParams p{};
    p.g = dz; p.x = x; p.dw = workspace;
    p.N = N; p.H = H; p.W = W; p.Cx = Cx; p.Cn = Cout;
    p.Ho = (H + 2 * pad - R) / stride + 1; p.Wo = (W + 2 * pad - S) / stride + 1;
    p.M = N * p.Ho * p.Wo; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
    p.ktiles = cdiv(p.M, 32);
    const long long ge = (long long)p.M * CoutG, xe = (long long)N * H * W * Cx;
    if (ge >= (1LL << 30) || xe >= (1LL << 30)) return fail(VATL_EINVAL, "conv2d_wgrad: a tensor exceeds 2^30 elements; split the batch");
    p.g_bytes = (unsigned)(ge * 4); p.x_bytes = (unsigned)(xe * 4);
    hipStream_t st = (hipStream_t)stream;
    const int64_t wsn = vatl_conv2d_wgrad_workspace_floats(Cout, Cin, R, S);
    if (hipMemsetAsync(workspace, 0, wsn * sizeof(float), st) != hipSuccess) return fail(VATL_ELAUNCH, "conv2d_wgrad: memset failed");
    int rc;
    WgradParams q = p;
    q.Gs = CoutG;                                         // row stride of dz; channels Cout..CoutG-1 are padding
    if (stem) rc = launch_wgrad<64, 32, 32, 32, true>(q, st);
    else if (Cout >= 128 && Cx >= 128) rc = launch_wgrad<128, 128, 64, 64, false>(q, st);
    else if (Cout >= 128) rc = launch_wgrad<128, 64, 64, 32, false>(q, st);
    else if (Cout > 32 && Cx >= 128) rc = launch_wgrad<64, 128, 32, 64, false>(q, st);
    else if (Cout > 32) rc = launch_wgrad<64, 64, 32, 32, false>(q, st);
    else rc = launch_wgrad<32, 128, 32, 32, false>(q, st);
    if (rc) return rc;
    const long long total = (long long)Cout * Cin * R * S;
    long long gsz = (total + 255) / 256; if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(unpack_conv_grad_kernel, dim3((unsigned)gsz), dim3(256), 0, st, workspace, dw, Cout, Cin, R, S, stem ? 8 : S, Cx);
    return check_launch("conv2d_wgrad");
}

extern "C" int64_t vatl_deconv4x4s2_wgrad_workspace_floats(int Cin, int Cout) { return (int64_t)Cin * 16 * Cout; }

// dw (Cin,Cout,4,4) of ConvTranspose2d(4,2,1): x NHWC (N,H,W,Cin) layer input, dy NHWC (N,2H,2W,Cout).
extern "C" int vatl_deconv4x4s2_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int H, int W, int Cin,
                                      int Cout, void* stream) {
    if (!x || !dy || !dw || !workspace) return fail(VATL_EINVAL, "deconv4x4s2_wgrad: null pointer");
    if ((Cin % 4) || (Cout % 4)) return fail(VATL_EINVAL, "deconv4x4s2_wgrad: channels must be multiples of 4");
    WgradParams p{};
    p.g = x; p.x = dy; p.dw = workspace;
    p.N = N; p.H = 2 * H; p.W = 2 * W; p.Cx = Cout; p.Cn = Cin;
    p.Gs = Cin;
    p.Ho = H; p.Wo = W; p.M = N * H * W; p.R = 4; p.S = 4; p.stride = 2; p.pad = 1;
    p.ktiles = cdiv(p.M, 32);
    const long long ge = (long long)p.M * Cin, xe = 4LL * p.M * Cout;
    if (ge >= (1LL << 30) || xe >= (1LL << 30)) return fail(VATL_EINVAL, "deconv4x4s2_wgrad: a tensor exceeds 2^30 elements; split the batch");
    p.g_bytes = (unsigned)(ge * 4); p.x_bytes = (unsigned)(xe * 4);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(workspace, 0, (size_t)Cin * 16 * Cout * sizeof(float), st) != hipSuccess) return fail(VATL_ELAUNCH, "deconv4x4s2_wgrad: memset failed");
    int rc;
    if (Cin >= 128 && Cout >= 128) rc = launch_wgrad<128, 128, 64, 64, false>(p, st);
    else if (Cin >= 128) rc = launch_wgrad<128, 64, 64, 32, false>(p, st);
    else if (Cout >= 128) rc = launch_wgrad<64, 128, 32, 64, false>(p, st);
    else rc = launch_wgrad<64, 64, 32, 32, false>(p, st);
    if (rc) return rc;
    const long long total = (long long)Cin * Cout * 16;
    long long gsz = (total + 255) / 256; if (gsz > 4096) gsz = 4096;
    hipLaunchKernelGGL(unpack_deconv_grad_kernel, dim3((unsigned)gsz), dim3(256), 0, st, workspace, dw, Cin, Cout);
    return check_launch("deconv4x4s2_wgrad");
}
