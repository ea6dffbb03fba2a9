// Weight gradient of Conv2d / ConvTranspose2d(4,2,1) as an implicit GEMM on the fp32 matrix cores.
//
//   dW[n][t][c] = sum_m G[m][n] * X[pix(m) + tap t][c]
//   G  (M, Cn)      "plain" operand, NHWC rows m = (b, oy, ox): the gradient of the conv output
//                   (for the transposed conv: the layer INPUT, Cn = Cin of the deconv)
//   X  (N,H,W,Cx)   "gathered" operand: the conv input (for the transposed conv: the output gradient)
//   t = (r, s)      filter tap, pix(m)+t = (oy*stride - pad + r, ox*stride - pad + s), zero outside
//
// GEMM shape: rows n (Cn), cols j = (t, c) (R*S*Cx), reduction over the M pixels — the long axis —
// so the launch is split along M: every split writes its partial tile into its own slice of a workspace
// (packed layout [Cn][R*S][Cx] per slice) and wgrad_reduce_kernel sums the slices in a fixed order while it
// restores the tensor layout — no atomics, bitwise reproducible.  Both operands are staged K-major ([m][n] / [m][c], the
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
    int d_oy, d_ox;      // (row, column) advance of a 32-pixel step: 32 = d_b*Ho*Wo + d_oy*Wo + d_ox
    int adv, adv_cx, adv_cy;   // element-offset advance of the gathered operand for that step / a column carry / a row carry
    unsigned g_bytes, x_bytes;
    int tpt;             // filter taps per block column tile (> 1 when Cx < BJ: a tile packs BJ / Cx taps instead of padding one tap's channels)
    int ablate;          // benchmarks only (vatl_tune_set(4, bits)): 1 = skip the epilogue (wrong results)
    long long slice;     // floats between the partial gradients of consecutive M-splits in the workspace
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
    static_assert(NWK == (STEM ? 2 : 1), "wgrad_cfg().nwk must match the waves that split the k-steps");
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
    // the filter tap this thread gathers for: the block's tap, or (packed tiles, Cx < BJ) tap group * tpt + its column's tap
    const int tpt = STEM ? 1 : p.tpt;
    const int cj0 = (tid % (BJ / 4)) * 4;
    const int my_tap = tpt > 1 ? tap * tpt + cj0 / p.Cx : tap;
    const bool tap_ok = STEM || my_tap < p.R * p.S;
    const int r = STEM ? tap : my_tap / p.S, s = STEM ? 0 : my_tap - r * p.S;
    const int kt0 = split * p.kt_per_split;
    const int kt1 = min(kt0 + p.kt_per_split, p.ktiles);
    if (kt0 >= kt1) return;

    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, p.g_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const int HoWo = p.Ho * p.Wo;

    // Operand addresses advance by 32 pixels per k-tile.  G (plain rows): one add, rows >= M fall outside the buffer
    // descriptor and read zeros.  X (gathered): byte-free element offset + (oy, ox) advanced with carries — the
    // straightforward form (two integer divisions and three multiplies per load) cost more VALU time than the
    // k-tile's MFMAs leave free.
    f32x4 rg[LG], rx[LX];
    unsigned goff[LG], gadv[LG];
#pragma unroll
    for (int q = 0; q < LG; ++q) {
        const int f = tid + 256 * q;
        const int row = f / (BN / 4), n = n0 + (f % (BN / 4)) * 4;
        const bool nv = n < p.Gs;
        goff[q] = nv ? (unsigned)((kt0 * 32 + row) * p.Gs + n) << 2 : WG_OOB;
        gadv[q] = nv ? (unsigned)(32 * p.Gs) << 2 : 0u;
    }
    int xm[LX], xoy[LX], xox[LX], xoff[LX];
    bool xcv[LX];
#pragma unroll
    for (int q = 0; q < LX; ++q) {
        const int f = tid + 256 * q;
        const int c4 = f % (BJ / 4);
        const int m = kt0 * 32 + f / (BJ / 4);
        const int b = m / HoWo;
        const int rem = m - b * HoWo;
        xm[q] = m;
        xoy[q] = rem / p.Wo;
        xox[q] = rem - xoy[q] * p.Wo;
        const int c = STEM ? 0 : (tpt > 1 ? (c4 * 4) % p.Cx : c0 + c4 * 4);
        xcv[q] = tap_ok && c < p.Cx;
        // STEM: float4 = one of the 8 taps of this filter row (4 channels): the column offset c4 rides in the offset
        xoff[q] = ((b * p.H + xoy[q] * p.stride - p.pad + r) * p.W + xox[q] * p.stride - p.pad + s + (STEM ? c4 : 0)) * p.Cx + c;
    }
    const int iyb = r - p.pad, ixb = s - p.pad;
    auto gload = [&]() {                                   // loads the next k-tile and advances the pixel state
#pragma unroll
        for (int q = 0; q < LG; ++q) {
            rg[q] = wg_load4(gr, goff[q]);
            goff[q] += gadv[q];
        }
#pragma unroll
        for (int q = 0; q < LX; ++q) {
            const int iy = xoy[q] * p.stride + iyb;
            const int ix = xox[q] * p.stride + ixb + (STEM ? (int)((tid + 256 * q) % (BJ / 4)) : 0);
            const bool ok = xcv[q] && xm[q] < p.M && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            rx[q] = wg_load4(xr, ok ? (unsigned)xoff[q] << 2 : WG_OOB);
            xm[q] += 32; xoy[q] += p.d_oy; xox[q] += p.d_ox; xoff[q] += p.adv;
            if (xox[q] >= p.Wo) { xox[q] -= p.Wo; ++xoy[q]; xoff[q] += p.adv_cx; }
            if (xoy[q] >= p.Ho) { xoy[q] -= p.Ho; xoff[q] += p.adv_cy; }
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

    gload();
    lstore(0);
    __syncthreads();
    const int half = lane >> 5, l31 = lane & 31;
    // k-loop: groups of SG k-steps (pixel pairs).  The next group's fragment reads, the next tile's global loads +
    // address arithmetic (first group) and the staging writes (last group) are issued in the shadow of the current
    // group's MFMAs (64 matrix-pipe cycles each), pinned with sched_group_barrier — same scheme as conv_igemm VAR2.
    constexpr int STEPS = 16 / NWK, SG = 4, NG = STEPS / SG, MPG = SG * TN * TJ;
    static_assert(NG >= 2, "k-loop needs at least two step groups");
    auto frag = [&](float (&a)[SG][TN], float (&b)[SG][TJ], int buf, int g) {
#pragma unroll
        for (int st = 0; st < SG; ++st) {
            const int row = 2 * ((g * SG + st) * NWK + wk) + half;    // pixel of this k-step handled by this lane half
#pragma unroll
            for (int i = 0; i < TN; ++i) a[st][i] = Gs[buf][row][wn * WN + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[st][j] = Xs[buf][row][wj * WJ + j * 32 + l31];
        }
    };
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        float a[2][SG][TN], b[2][SG][TJ];
        frag(a[0], b[0], buf, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) frag(a[(g + 1) & 1], b[(g + 1) & 1], buf, g + 1);
            if (g == 0) gload();                           // tail: one harmless extra tile, keeps the body branch-free
            if (g == NG - 1) lstore(buf ^ 1);
#pragma unroll
            for (int st = 0; st < SG; ++st)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][st][i], b[g & 1][st][j], acc[i][j], 0, 0, 0);
            if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, SG * (TN + TJ), 0);
            if (g == 0) {
#pragma unroll
                for (int q = 0; q < MPG; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x216, MPG >= 16 ? 6 : 12, 0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < MPG; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x216, 2, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

#ifdef VATL_ABLATION
    if (p.ablate & 1) return;          // profiling build only
#endif
    // D[row = (e&3) + 8*(e>>2) + 4*(lane>>5)] -> n,  [col = lane&31] -> channel: coalesced stores of this split's partial
    // tile into its own slice of the workspace (no atomics: the reduction over splits runs in a fixed order afterwards)
    const int Kp = (STEM ? p.R * 8 : p.R * p.S) * p.Cx;
    float* __restrict__ part = p.dw + ((long long)split * NWK + wk) * p.slice;      // waves that share a tile and split its k-steps own a slice each
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int cj = wj * WJ + j * 32 + l31;                    // column inside the block tile
            const int col = STEM ? tap * 32 + cj : (tpt > 1 ? tap * tpt * p.Cx + cj : tap * p.Cx + c0 + cj);   // packed tiles: taps are contiguous in the layout
            const bool cv = STEM ? true : (tpt > 1 ? col < Kp : (c0 + cj) < p.Cx);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn * WN + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * half;
                if (cv && n < p.Cn) part[(long long)n * Kp + col] = acc[i][j][e];
            }
        }
}

static std::atomic<int> g_wgrad_blocks{0};    // 0 = by filter size (below); vatl_tune_set(3, v) forces a target
static std::atomic<int> g_wgrad_ablate{0};   // vatl_tune_set(3, v)

// tile configuration of a weight-gradient GEMM: rows n (Cn), columns (tap, channel of Cx)
struct WgradCfg { int bn, bj; bool stem; int nwk; };      // nwk: waves of a block that split the k-steps of one tile (one slice each)
static WgradCfg wgrad_cfg(int Cn, int Cx, bool stem) {
    if (stem) return {64, 32, true, 2};
    if (Cn >= 128 && Cx >= 128) return {128, 128, false, 1};
    if (Cn >= 128) return {128, 64, false, 1};
    if (Cn > 32 && Cx >= 128) return {64, 128, false, 1};
    if (Cn > 32) return {64, 64, false, 1};
    return {32, 128, false, 1};
}

// number of M-splits: at most ~target blocks (two per CU resident: 512 per wave), at least 4 k-tiles each.  Measured at
// B = 120 (tools/wgrad_bench.py): 1x1 filters are ~10 % faster with one 512-block wave (half the partial slices to
// write and sum), 3x3 filters want two waves.
struct WgradPlan { int tiles, splits, kt_per_split, slices; };   // slices = splits * nwk partial gradients to sum
// taps per column tile: narrow inputs (Cx < BJ, HRNet's 32- / 64-channel branches) pack BJ / Cx taps into a tile instead of
// multiplying zero columns (a 32-channel 3x3 filter: 3 tiles of 4 taps instead of 9 tiles that are 3/4 padding)
static int wgrad_tpt(const WgradCfg& c, int Cx) { return (!c.stem && Cx < c.bj && c.bj % Cx == 0) ? c.bj / Cx : 1; }
static WgradPlan wgrad_plan(const WgradCfg& c, int Cn, int Cx, int R, int S, long long M) {
    const int tpt = wgrad_tpt(c, Cx);
    const int n_tiles = cdiv(Cn, c.bn), taps = c.stem ? R : cdiv(R * S, tpt), j_tiles = c.stem ? 1 : cdiv(Cx, c.bj);
    const int tiles = n_tiles * taps * j_tiles;
    const int ktiles = cdiv(M, 32);
    int target = g_wgrad_blocks.load(std::memory_order_relaxed);
    if (target <= 0) target = (R * S == 1) ? 512 : 1024;
    int splits = target / tiles;                                              // floor: stay within a whole number of 512-block waves
    const int max_splits = (ktiles + 3) / 4;
    if (splits > max_splits) splits = max_splits;
    if (splits < 1) splits = 1;
    const int kps = (ktiles + splits - 1) / splits;
    const int sp = (ktiles + kps - 1) / kps;                                  // every split owns >= 1 k-tile
    return {tiles, sp, kps, sp * c.nwk};
}

template <int BN, int BJ, int WN, int WJ, bool STEM>
static int launch_wgrad_t(WgradParams p, const WgradPlan& plan, hipStream_t st) {
    p.n_tiles = cdiv(p.Cn, BN);
    p.tpt = (!STEM && p.Cx < BJ && BJ % p.Cx == 0) ? BJ / p.Cx : 1;
    p.taps = STEM ? p.R : cdiv(p.R * p.S, p.tpt);
    p.j_tiles_per_tap = STEM ? 1 : cdiv(p.Cx, BJ);
    const int hw = p.Ho * p.Wo, rem = 32 % hw;
    p.d_oy = rem / p.Wo; p.d_ox = rem % p.Wo;
    p.adv = ((32 / hw) * p.H * p.W + p.d_oy * p.stride * p.W + p.d_ox * p.stride) * p.Cx;
    p.adv_cx = (p.stride * p.W - p.Wo * p.stride) * p.Cx;
    p.adv_cy = (p.H * p.W - p.Ho * p.stride * p.W) * p.Cx;
#ifdef VATL_ABLATION
    p.ablate = g_wgrad_ablate.load(std::memory_order_relaxed);
#else
    p.ablate = 0;
#endif
    p.kt_per_split = plan.kt_per_split;
    hipLaunchKernelGGL((conv_wgrad_kernel<BN, BJ, WN, WJ, STEM>), dim3((unsigned)(plan.tiles * plan.splits)), dim3(256), 0, st, p);
    meter_add(0, 2.0 * (double)plan.tiles * BN * BJ * (double)cdiv(p.M, 32) * 32.0);
    meter_route(kRouteWgrad);
    return check_launch("conv_wgrad");
}

static int launch_wgrad(const WgradParams& p, const WgradCfg& c, const WgradPlan& plan, hipStream_t st) {
    if (c.stem) return launch_wgrad_t<64, 32, 32, 32, true>(p, plan, st);
    if (c.bn == 128 && c.bj == 128) return launch_wgrad_t<128, 128, 64, 64, false>(p, plan, st);
    if (c.bn == 128) return launch_wgrad_t<128, 64, 64, 32, false>(p, plan, st);
    if (c.bn == 64 && c.bj == 128) return launch_wgrad_t<64, 128, 32, 64, false>(p, plan, st);
    if (c.bn == 64) return launch_wgrad_t<64, 64, 32, 32, false>(p, plan, st);
    return launch_wgrad_t<32, 128, 32, 32, false>(p, plan, st);
}

// Sum of the per-split partial gradients (fixed order: eight split lanes accumulate strided splits, four interleaved chains each,
// then the lanes are added in order) while the packed layout goes back to the tensor's own layout.  Threads run along the PACKED
// index so that every slice is read with coalesced 128-byte rows; block = 32 packed elements x 8 split lanes.
//   MODE 0: conv     packed [Cout][R][Spad][CinPad] -> out (Cout,Cin,R,S)
//   MODE 1: deconv   packed [Cin][16][Cout]         -> out (Cin,Cout,4,4)
template <int MODE>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ packed, float* __restrict__ out, int A, int B, int R, int S,
                                                           int Spad, int CinPad, int slices, long long slice, long long packed_total) {
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, q = threadIdx.x >> 5;
    const long long j = (long long)blockIdx.x * 32 + e;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;      // four independent chains: the loads of one lane overlap
    if (j < packed_total) {
        const float* src = packed + j;
        int k = q;
        for (; k + 24 < slices; k += 32) {
            a0 += src[(long long)k * slice]; a1 += src[(long long)(k + 8) * slice];
            a2 += src[(long long)(k + 16) * slice]; a3 += src[(long long)(k + 24) * slice];
        }
        for (; k < slices; k += 8) a0 += src[(long long)k * slice];
    }
    part[q][e] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q != 0 || j >= packed_total) return;
    float v = part[0][e];
#pragma unroll
    for (int t = 1; t < 8; ++t) v += part[t][e];
    if (MODE == 0) {                                   // A = Cout, B = Cin
        const int c = (int)(j % CinPad);
        long long t = j / CinPad;
        const int s = (int)(t % Spad); t /= Spad;
        const int r = (int)(t % R);
        const int o = (int)(t / R);
        if (c < B && s < S) out[(((long long)o * B + c) * R + r) * S + s] = v;
    } else {                                           // A = Cin, B = Cout
        const int o = (int)(j % B);
        long long t = j / B;
        const int k = (int)(t % 16);
        const int c = (int)(t / 16);
        out[((long long)c * B + o) * 16 + k] = v;
    }
}

}  // namespace vatl

using namespace vatl;

namespace vatl {
__attribute__((visibility("hidden"))) int tune_wgrad_blocks(int blocks) {       // reached through vatl_tune_set(3, blocks) / (4, bits) of the profiling variant only
    if (blocks < 0) { g_wgrad_ablate.store(-blocks - 1, std::memory_order_relaxed); return 0; }
    if (blocks < 64 || blocks > 65536) return -1;
    g_wgrad_blocks.store(blocks, std::memory_order_relaxed);
    return 0;
}
}  // namespace vatl

static int64_t conv_packed_floats(int Cout, int Cin, int R, int S) { return Cin == 3 ? (int64_t)Cout * R * 8 * 4 : (int64_t)Cout * R * S * Cin; }

extern "C" int64_t vatl_conv2d_wgrad_workspace_floats(int Cout, int Cin, int R, int S, int64_t M) {
    const bool stem = (Cin == 3);
    const int Cx = stem ? 4 : Cin;
    const WgradPlan plan = wgrad_plan(wgrad_cfg(Cout, Cx, stem), Cout, Cx, R, S, M);
    return conv_packed_floats(Cout, Cin, R, S) * plan.slices;
}

// dw (Cout,Cin,R,S) = sum over pixels of dz (x) gathered input.  x NHWC (N,H,W,Cin) (Cin = 4 padded for the
// 3-channel stem), dz NHWC (N,Ho,Wo,CoutG) where CoutG >= Cout is the channel stride of dz.  The pixel range is split
// over blocks; every split writes its partial gradient into its own slice of the workspace and one pass sums the slices
// in split order while it restores the OIHW layout: no atomics, bitwise reproducible.
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
    p.Gs = CoutG;                                         // row stride of dz; channels Cout..CoutG-1 are padding
    const long long ge = (long long)p.M * CoutG, xe = (long long)N * H * W * Cx;
    if (ge + 64LL * CoutG >= (1LL << 30) || xe >= (1LL << 30)) return fail(VATL_EINVAL, "conv2d_wgrad: a tensor exceeds 2^30 elements; split the batch");
    p.g_bytes = (unsigned)(ge * 4); p.x_bytes = (unsigned)(xe * 4);
    hipStream_t st = (hipStream_t)stream;
    const WgradCfg cfg = wgrad_cfg(Cout, Cx, stem);
    const WgradPlan plan = wgrad_plan(cfg, Cout, Cx, R, S, p.M);
    p.slice = conv_packed_floats(Cout, Cin, R, S);
    const int rc = launch_wgrad(p, cfg, plan, st);
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel<0>, dim3((unsigned)((p.slice + 31) / 32)), dim3(256), 0, st, workspace, dw, Cout, Cin, R, S, stem ? 8 : S, Cx,
                       plan.slices, p.slice, p.slice);
    return check_launch("conv2d_wgrad");
}

extern "C" int64_t vatl_deconv4x4s2_wgrad_workspace_floats(int Cin, int Cout, int64_t M) {
    const WgradPlan plan = wgrad_plan(wgrad_cfg(Cin, Cout, false), Cin, Cout, 4, 4, M);
    return (int64_t)Cin * 16 * Cout * plan.slices;
}

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
    if (ge + 64LL * Cin >= (1LL << 30) || xe >= (1LL << 30)) return fail(VATL_EINVAL, "deconv4x4s2_wgrad: a tensor exceeds 2^30 elements; split the batch");
    p.g_bytes = (unsigned)(ge * 4); p.x_bytes = (unsigned)(xe * 4);
    hipStream_t st = (hipStream_t)stream;
    const WgradCfg cfg = wgrad_cfg(Cin, Cout, false);
    const WgradPlan plan = wgrad_plan(cfg, Cin, Cout, 4, 4, p.M);
    p.slice = (long long)Cin * 16 * Cout;
    const int rc = launch_wgrad(p, cfg, plan, st);
    if (rc) return rc;
    hipLaunchKernelGGL(wgrad_reduce_kernel<1>, dim3((unsigned)((p.slice + 31) / 32)), dim3(256), 0, st, workspace, dw, Cin, Cout, 4, 4, 4, Cout,
                       plan.slices, p.slice, p.slice);
    return check_launch("deconv4x4s2_wgrad");
}
