// ResNet stem for inference in ONE kernel:  NCHW crops -> conv 7x7 / stride 2 / pad 3 (3 -> 64) -> folded BatchNorm -> ReLU -> max-pool 3x3 / stride 2 / pad 1
// -> NHWC activations of the first bottleneck (Resnet.py:155-158, 171-172).  Replaces three launches of the inference plan (vatl_nchw_to_nhwc, the
// implicit-GEMM stem, vatl_maxpool3x3s2_fwd) and their 3.2 GB stem activation that was written and read back per 1024 crops.
//
// Why a kernel of its own.  As an implicit GEMM over 4-channel pixels the stem multiplies K = 7 rows x 8 taps x 4 channels = 224 for 147 real
// products, and its 64-pixel tiles are runs of a row-major pixel index, so the 3x3 pooling windows straddle tiles.  Here:
//   * K = 3 channels x 7 rows x 8 taps (7 + 1 zero) = 168: the input stays PLANAR (the reference's NCHW, read directly), each input row is kept in LDS
//     split into its even and odd pixels, so that for a fixed (channel, filter row) the four MFMA k-pairs of an output pixel are (odd tap i, even tap i),
//     i = 0..3 — lane half 0 reads O[ox + i - 2], lane half 1 reads E[ox + i - 1]: consecutive lanes read consecutive floats (conflict-free ds_read_b32),
//     one read per MFMA; the filter (84 values per lane and 32 output channels) lives in registers.
//   * a block owns an image (or a band of its rows) and slides down it two stem rows per step; waves (row, 32-channel half) each compute one full
//     stem row (W / 2 = 32 TPR pixels = TPR MFMA row blocks), pool it horizontally through a small wave-private LDS tile, the two rows meet in LDS for
//     the vertical maximum with the odd row carried over from the step before.  Input rows arrive four per step through registers while the MFMAs run.
// Post-ReLU values are >= 0, so 0 is the neutral element of every maximum (image borders).
// Per-pixel arithmetic depends on the image's own pixels only: results are independent of the batch position.  The summation order over K differs
// from the implicit-GEMM stem (channel-major instead of row-major): same precision class, not the same bits — the training path keeps the old stem.
#include "common.h"

namespace vatl {

constexpr int SP_ZREG = 512;                     // floats of the zero region (reads of the padded tap land here whatever their immediate offset)

// KS = 7 (pad 3, + max-pool): the ResNet stem.  KS = 3 (pad 1, no pooling): HRNet's first stem conv (hrnet.py:109-110, 426-428) — the same sliding block
// with K = 3 channels x 3 rows x 4 taps (3 + 1 zero) = 36 instead of the 96 of the 4-channel implicit GEMM, writing the stem rows themselves.
template <int TPR, int KS> struct StemPoolLds {
    static constexpr int RING = KS + 6;          // input rows resident: KS + 2 in use (two stem rows) + 4 arriving
    static constexpr int WO = 32 * TPR;          // stem output columns
    static constexpr int IW = WO + 4;            // entries of an even / odd pixel array: index q + 2 <-> pixel 2q (even) / 2q + 1 (odd); two zeros on either side
    static constexpr int ROW = 3 * 2 * IW;       // floats of one ring slot: [channel][parity][IW]
    static constexpr int IN = RING * ROW;
    static constexpr int STG = 33 * 32;          // per wave: [1 + 32 pixels][32 channels]
    static constexpr int HP = 2 * (WO / 2) * 64; // horizontally pooled rows of this step: [stem row parity][pooled column][64 channels]
    static constexpr int FLOATS = SP_ZREG + IN + (KS == 7 ? 4 * STG + HP : 0);
};

struct StemPoolParams {
    const float* x;          // (N, 3, H, W) fp32
    const float* w;          // packed: [2 channel halves][84][64 lanes]
    const float* scale;      // folded BatchNorm (64)
    const float* bias;
    float* y;                // (N, H / 4, W / 4, 64)
    int N, H, W;
    int bands, steps_per_band;   // a block = (image, band of pooled rows)
};

template <int TPR, int KS>
__global__ __launch_bounds__(256, 2) void stem_pool_kernel(StemPoolParams p) {
    using L = StemPoolLds<TPR, KS>;
    constexpr int WO = L::WO, IW = L::IW, ROW = L::ROW, PW = WO / 2, SP_RING = L::RING;
    constexpr bool POOL = KS == 7;
    constexpr int PAD = KS / 2, NPAIR = (KS + 1) / 2, NW = 3 * KS * NPAIR;   // k-pairs per (channel, filter row); filter values per lane
    constexpr int OO = 2 - (PAD + 1) / 2, EO = 2 - (PAD - 1) / 2;           // pair i: half 0 reads the odd array at ox + i + OO, half 1 the even array at ox + i + EO
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Zr = smem;                            // zeros
    float* In = smem + SP_ZREG;
    float* Stg = In + L::IN;
    float* Hp = Stg + 4 * L::STG;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nh = wave & 1, rsel = wave >> 1;   // this wave: channels [32 nh, 32 nh + 32) of stem row 2 i + rsel
    const int m = lane & 31, h = lane >> 5;
    const int b = blockIdx.x / p.bands, band = blockIdx.x - b * p.bands;
    const int HO = p.H >> 1, PH = p.H >> 2;      // stem rows, pooled rows
    const int i_first = band * p.steps_per_band, i_end = min(i_first + p.steps_per_band, PH);
    if (i_first >= i_end) return;
    const int i_begin = (POOL && i_first > 0) ? i_first - 1 : i_first;   // a pooling band that does not start at the top first recomputes the odd stem row above it (results not stored)

    // ---- filter fragments of this wave's 32 channels: 84 k-pairs, lane half 0 = the odd-pixel taps (kx = 2 i), half 1 = the even-pixel taps (kx = 2 i + 1; i = 3: zero)
    float wr[NW];
    {
        const float* wp = p.w + (long long)nh * NW * 64 + lane;
#pragma unroll
        for (int k = 0; k < NW; ++k) wr[k] = wp[k * 64];
    }
    const int ch = 32 * nh + m;                  // (as MFMA column: this lane's output channel)
    const float sc = p.scale[ch], bi = p.bias[ch];

    // ---- LDS init: zero region, and the halo entries of every ring slot (never written again)
    for (int k = tid; k < SP_ZREG; k += 256) Zr[k] = 0.f;
    for (int k = tid; k < SP_RING * 6 * 4; k += 256) {
        const int arr = k >> 2, e = k & 3;
        In[arr * IW + (e < 2 ? e : IW - 4 + e)] = 0.f;
    }

    // ---- input rows: global (planar, W floats per row) -> registers -> LDS (even / odd split).  A thread moves float2 pieces: piece id = ((r * 3 + c) * WO + q)
    constexpr int PIECES4 = 4 * 3 * WO;          // four rows per step
    constexpr int NP4 = (PIECES4 + 255) / 256;
    const float* xb = p.x + (long long)b * 3 * p.H * p.W;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto load_rows = [&](f32x2 (&regs)[NP4], int row0, int nrows) {      // rows row0 .. row0 + nrows - 1 (nrows <= 4)
#pragma unroll
        for (int u = 0; u < NP4; ++u) {
            const int id = tid + 256 * u;
            const int r = id / (3 * WO), rem = id - r * (3 * WO);
            const int c = rem / WO, q = rem - c * WO;
            const int row = row0 + r;
            f32x2 v = {0.f, 0.f};
            if (id < PIECES4 && r < nrows && (unsigned)row < (unsigned)p.H) v = *reinterpret_cast<const f32x2*>(xb + ((long long)c * p.H + row) * p.W + 2 * q);
            regs[u] = v;
        }
    };
    auto store_rows = [&](const f32x2 (&regs)[NP4], int row0, int nrows) {
#pragma unroll
        for (int u = 0; u < NP4; ++u) {
            const int id = tid + 256 * u;
            const int r = id / (3 * WO), rem = id - r * (3 * WO);
            const int c = rem / WO, q = rem - c * WO;
            if (id < PIECES4 && r < nrows) {
                int slot = (row0 + r) % SP_RING; if (slot < 0) slot += SP_RING;
                float* dst = In + slot * ROW + c * 2 * IW + q + 2;
                dst[0] = regs[u][0];             // even pixel 2 q
                dst[IW] = regs[u][1];            // odd pixel 2 q + 1
            }
        }
    };
    f32x2 rg[NP4];
    // prologue: rows 4 i_begin - PAD .. 4 i_begin + 1 + KS - PAD (KS + 2 rows)
    for (int r0 = 4 * i_begin - PAD; r0 < 4 * i_begin + 2 + KS - PAD; r0 += 4) {
        const int nr = min(4, 4 * i_begin + 2 + KS - PAD - r0);
        load_rows(rg, r0, nr);
        store_rows(rg, r0, nr);
    }
    __syncthreads();

    // per-lane read base inside a ring slot: half 0 reads the ODD array at 32 t + m + i + OO, half 1 the EVEN array at 32 t + m + i + EO
    const int lbase = (h == 0 ? IW + OO : EO) + m;
    f32x4 carry[3];                              // horizontally pooled odd stem row of the step before: this thread's (pooled column, 4 channels) x 3
#pragma unroll
    for (int k = 0; k < 3; ++k) carry[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    float* stg = Stg + wave * L::STG;

    for (int i = i_begin; i < i_end; ++i) {
        // rows of the next step, in flight while this one is multiplied
        const bool more = i + 1 < i_end;
        if (more) load_rows(rg, 4 * i + 2 + KS - PAD, 4);
        const int oy = 2 * i + rsel;             // this wave's stem row
        int rb[KS], rbp[KS];                     // float index of (filter row ky, channel 0, tile 0, pair 0) for this lane; rbp: the same for the padded pair (half 1 -> zero region)
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            int slot = (2 * oy + ky - PAD) % SP_RING; if (slot < 0) slot += SP_RING;
            rb[ky] = SP_ZREG + slot * ROW + lbase;
            rbp[ky] = h ? 0 : rb[ky];
        }
        f32x16 acc[TPR];
#pragma unroll
        for (int t = 0; t < TPR; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int ky = 0; ky < KS; ++ky)
#pragma unroll
                for (int t = 0; t < TPR; ++t) {
                    const int o = c * 2 * IW + 32 * t;
#pragma unroll
                    for (int k = 0; k < NPAIR - 1; ++k)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(smem[rb[ky] + o + k], wr[(c * KS + ky) * NPAIR + k], acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(smem[rbp[ky] + o + NPAIR - 1], wr[(c * KS + ky) * NPAIR + NPAIR - 1], acc[t], 0, 0, 0);
                }

        if constexpr (!POOL) {
            // ---- folded BatchNorm + ReLU, stored as the stem row itself: (N, H / 2, W / 2, 64); a lane's 32 neighbours write 128 contiguous bytes of a pixel
            float* yrow = p.y + (((long long)b * HO + oy) * WO) * 64 + ch;
#pragma unroll
            for (int t = 0; t < TPR; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int px = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * h;
                    yrow[px * 64] = fmaxf(fmaf(acc[t][e], sc, bi), 0.f);
                }
            __syncthreads();                     // every wave is done reading this step's input rows
            if (more) store_rows(rg, 4 * i + 2 + KS - PAD, 4);
            __syncthreads();
            continue;
        }

        // ---- folded BatchNorm + ReLU, horizontal 3-window maximum with stride 2 through the wave's LDS tile (row 0 = the pixel left of the tile)
        if (lane < 32) stg[lane] = 0.f;          // left of the image: neutral
#pragma unroll
        for (int t = 0; t < TPR; ++t) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int px = (e & 3) + 8 * (e >> 2) + 4 * h;
                stg[(1 + px) * 32 + m] = fmaxf(fmaf(acc[t][e], sc, bi), 0.f);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            float hpv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = h + 2 * j;         // pooled column inside the tile: pixels 2 q - 1, 2 q, 2 q + 1 = tile rows 2 q, 2 q + 1, 2 q + 2
                hpv[j] = fmaxf(fmaxf(stg[(2 * q) * 32 + m], stg[(2 * q + 1) * 32 + m]), stg[(2 * q + 2) * 32 + m]);
            }
            const float last = stg[32 * 32 + m];  // pixel 31: left neighbour of the next tile
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < 32) stg[lane] = last;
#pragma unroll
            for (int j = 0; j < 8; ++j) Hp[(rsel * PW + 16 * t + h + 2 * j) * 64 + ch] = hpv[j];
        }
        __syncthreads();                         // both stem rows of the step are in Hp; every wave is done reading this step's input rows

        // ---- vertical maximum (odd row of the step before, even row, odd row) and store; the next step's input rows go to LDS
        if (more) store_rows(rg, 4 * i + 2 + KS - PAD, 4);
        {
            constexpr int V4 = PW * 16;          // float4 pieces of one pooled row
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int id = tid + 256 * k;
                if (id < V4) {
                    const f32x4 ev = *reinterpret_cast<const f32x4*>(Hp + id * 4);
                    const f32x4 od = *reinterpret_cast<const f32x4*>(Hp + PW * 64 + id * 4);
                    f32x4 o;
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) o[c4] = fmaxf(fmaxf(carry[k][c4], ev[c4]), od[c4]);
                    carry[k] = od;
                    if (i >= i_first) *reinterpret_cast<f32x4*>(p.y + (((long long)b * PH + i) * PW) * 64 + id * 4) = o;
                }
            }
        }
        __syncthreads();                         // Hp is free again, the new input rows are visible
    }
    (void)HO;
}

}  // namespace vatl

using namespace vatl;

extern "C" int64_t vatl_stem_pool_weight_floats(void) { return 2 * 84 * 64; }
extern "C" int64_t vatl_stem3_weight_floats(void) { return 2 * 18 * 64; }

// w (64, 3, KS, KS) OIHW -> [channel half][k = (c * KS + ky) * NPAIR + i][lane = 32 half + n]: half 0 = w[n][c][ky][2 i], half 1 = w[n][c][ky][2 i + 1] (beyond the filter: 0)
__global__ void stem_pool_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int KS) {
    const int NPAIR = (KS + 1) / 2, NW = 3 * KS * NPAIR;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * NW * 64) return;
    const int lane = idx & 63, k = (idx >> 6) % NW, nhalf = idx / (NW * 64);
    const int i = k % NPAIR, cky = k / NPAIR, c = cky / KS, ky = cky - KS * c;
    const int n = 32 * nhalf + (lane & 31), hh = lane >> 5;
    const int kx = 2 * i + hh;
    out[idx] = kx < KS ? w[((n * 3 + c) * KS + ky) * KS + kx] : 0.f;
}

extern "C" int vatl_pack_stem_pool_weight(const float* w_oihw, float* packed, void* stream) {
    if (!w_oihw || !packed) return fail(VATL_EINVAL, "pack_stem_pool_weight: null pointer");
    hipLaunchKernelGGL(stem_pool_pack_kernel, dim3((2 * 84 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, 7);
    return check_launch("stem_pool_pack");
}

extern "C" int vatl_pack_stem3_weight(const float* w_oihw, float* packed, void* stream) {
    if (!w_oihw || !packed) return fail(VATL_EINVAL, "pack_stem3_weight: null pointer");
    hipLaunchKernelGGL(stem_pool_pack_kernel, dim3((2 * 18 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, packed, 3);
    return check_launch("stem3_pack");
}

// 1 when the fused kernels serve this input size (W / 2 a multiple of 32 up to 96 columns, H a multiple of 4), else 0 (the caller keeps the generic path)
extern "C" int vatl_stem_pool_supported(int H, int W) {
    const int wo = W / 2;
    return (H > 0 && W > 0 && (H & 3) == 0 && (W & 3) == 0 && wo % 32 == 0 && wo >= 32 && wo <= 96) ? 1 : 0;
}

template <int TPR, int KS>
static int launch_stem_pool(const StemPoolParams& p, hipStream_t st) {
    auto kern = stem_pool_kernel<TPR, KS>;
    static std::atomic<unsigned> configured{0};
    constexpr int smem = StemPoolLds<TPR, KS>::FLOATS * (int)sizeof(float);
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, configured, "stem_pool")) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.N * p.bands)), dim3(256), smem, st, p);
    // executed MFMA FLOPs: every step multiplies 2 stem rows x (W / 2) pixels x 64 channels x K = 3 KS (KS + 1) (+ one recomputed step per pooling band below the first)
    const double steps = (double)p.N * ((p.H >> 2) + (KS == 7 ? p.bands - 1 : 0));
    meter_add(0, 2.0 * steps * 2.0 * (p.W / 2) * 64.0 * (3.0 * KS * (KS + 1)));
    meter_route(kRouteStemPool);
    return check_launch("stem_pool");
}

template <int KS>
static int stem_impl(const float* x_nchw, const float* w_packed, const float* scale, const float* bias, float* y_nhwc, int N, int H, int W, void* stream, const char* what) {
    if (N <= 0) return 0;
    if (!x_nchw || !w_packed || !scale || !bias || !y_nhwc) return fail(VATL_EINVAL, "%s: null pointer", what);
    if (!vatl_stem_pool_supported(H, W)) return fail(VATL_EINVAL, "%s: %dx%d input not served by the fused kernel (W / 2 must be 32, 64 or 96, H %% 4 == 0)", what, H, W);
    if ((((uintptr_t)x_nchw) & 7) != 0) return fail(VATL_EINVAL, "%s: input must be 8-byte aligned", what);
    StemPoolParams p{};
    p.x = x_nchw; p.w = w_packed; p.scale = scale; p.bias = bias; p.y = y_nhwc; p.N = N; p.H = H; p.W = W;
    // one block per image once the images alone fill the 512 block slots; fewer images are cut into bands of row pairs (a pooling band re-computes one stem row)
    const int PH = H >> 2;
    int bands = 1;
    while ((long long)N * bands < 512 && bands * 2 <= PH / 4) bands *= 2;
    p.bands = bands; p.steps_per_band = (PH + bands - 1) / bands;
    hipStream_t st = (hipStream_t)stream;
    const int tpr = W / 64;
    if (tpr == 1) return launch_stem_pool<1, KS>(p, st);
    if (tpr == 2) return launch_stem_pool<2, KS>(p, st);
    return launch_stem_pool<3, KS>(p, st);
}

extern "C" int vatl_stem7x7s2_pool_fwd(const float* x_nchw, const float* w_packed, const float* scale, const float* bias, float* y_nhwc, int N, int H, int W,
                                       void* stream) {
    return stem_impl<7>(x_nchw, w_packed, scale, bias, y_nhwc, N, H, W, stream, "stem7x7s2_pool_fwd");
}

// conv 3x3 / stride 2 / pad 1 (3 -> 64) + folded BatchNorm + ReLU from NCHW crops to NHWC (N, H / 2, W / 2, 64): HRNet's conv1 + bn1 + relu (hrnet.py:109-110, 426-428)
extern "C" int vatl_stem3x3s2_fwd(const float* x_nchw, const float* w_packed, const float* scale, const float* bias, float* y_nhwc, int N, int H, int W, void* stream) {
    return stem_impl<3>(x_nchw, w_packed, scale, bias, y_nhwc, N, H, W, stream, "stem3x3s2_fwd");
}
