// 3x3 / stride 1 / pad 1 convolution as Winograd F(2x2, 3x3) on the gfx950 fp32 matrix cores.
//
// Why.  The 3x3 stride-1 layers are a quarter of the R50 trunk's time and nearly all of HRNet's (profiles/r03_layer_report.txt), and the
// implicit GEMM already runs them at 76 - 89 % of the fp32 MFMA peak: the only way to make them substantially faster is fewer multiplies.
// F(2x2, 3x3) computes a 2x2 output tile from its 4x4 input tile with 16 multiplies per (input channel, output channel) instead of 36:
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          U = G g G^T (16 values per filter, packed once: vatl_pack_winograd_weight)
//                                                   V = B^T d B (16 values per tile and input channel, computed on the fly)
//
// i.e. 16 independent GEMMs  M_p[tile][n] = sum_c V_p[tile][c] U_p[c][n]  (p = (xi, nu), the position in the 4x4 transform domain), 2.25x
// fewer MFMAs than the implicit GEMM, still exact fp32 products with fp32 accumulation (cuDNN picks the same algorithm for these layers
// of the reference: the rounding differs from the direct sum in the last bits, not in class; tests hold it to the same tolerance).
//
// Block = 64 tiles (flat index over image, tile row, tile column) x BN = 32 NH output channels x all 16 positions; 4 NH waves.  Wave
// (xi, nh) owns the four positions (xi, nu = 0..3) for both 32-tile halves and its 32 channels: 8 accumulator tiles = 128 registers.
//   * B operand (U): packed in MFMA fragment order, a wave's ds-free 16-byte-per-lane load is 1 KB contiguous; straight from L2 into
//     registers one 8-channel step ahead (no LDS: every wave needs a different slice).
//   * A operand (V): the raw input pixels of the block's tiles are staged in LDS 16 channels at a time (double buffer, register staging one
//     stage ahead), de-duplicated along the tile row: per input row i of the tiles, the even (x = 2 tx) and odd (x = 2 tx + 1) pixels of
//     64 consecutive tiles + one halo slot each; tile column j of tile t is then slot t or t + 1 of the odd / even array, and the left /
//     right image border is a lane's address pointing at a zero pixel.  A wave needs two of the four tile rows (row transform: d0 - d2,
//     d1 + d2, d2 - d1, d1 - d3 for xi = 0..3) and forms its four V_p = column transform in registers: 8 ds_read_b128 and 8 vector adds
//     per 16 MFMAs.  The 16-byte read of a lane is four consecutive channels = four MFMA k-steps (the same k permutation on both operands).
//   * Output transform: the nu sum happens in registers (4 accumulator tiles -> 2), the xi sum through LDS (the staging buffers are free
//     by then): every wave writes its two partial tiles, every thread then combines the four xi for whole 16-byte channel groups and
//     stores full NHWC rows, with scale / bias / residual / ReLU (and the BatchNorm statistics of the training forward) fused.
// The per-output arithmetic depends only on the tile's own pixels: results are independent of the batch position (SURVEY.md §7 hard part 3).
#include "common.h"
#include "winograd_pack.h"

#include <algorithm>
#include <atomic>

namespace vatl {

struct WinoParams {
    const float* x;
    const float* u;
    const float* scale;
    const float* bias;
    const float* res;
    float* y;
    double* stats;
    // BatchNorm-backward fusion (data-gradient launches of the fine-tune step; semantics of ConvParams::bz.. in conv_igemm.hip): the
    // tile being stored is dL/dy of a Conv+BN(+ReLU) layer whose conv output is bz; the epilogue applies that layer's ReLU mask, stores
    // the masked gradient g and accumulates (sum g, sum g*xhat) into `stats`
    const float* bz;
    const float* bmy;
    const float* bsc;
    const float* bbi;
    const float* bmu;
    const float* bis;
    int N, H, W, Cin, Cout;
    int TH, TW, tpi, Mtiles;          // tiles per image column / row / image, tiles in the launch
    int m_tiles, n_tiles;
    int relu;
    int stages;                       // Cin / 16
    int nhp;                          // 32-channel groups per filter tile of the packing (vatl_pack_winograd_weight: 1 if Cout <= 32, else 2)
    int ablate;                       // profiling library only (vatl_tune_set(17, bits), wrong results): 1 no output transform, 2 no LDS
                                      // reads / input transform, 4 no filter loads, 8 no staging DMA, 16 no barriers
    unsigned x_bytes, u_bytes, y_bytes;
};

constexpr unsigned WOOB = 0xFFFFFFFFu;
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 wbuf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ void wbuf_store4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wu32x4, v), r, byte_off, 0, 0);
}

constexpr int W_CK = 16;                        // channels per LDS stage
constexpr int W_ZERO = 16;                      // floats of the zero pixel in front of the stages
template <int HALVES>
struct WinoGeom {
    static constexpr int TB = 32 * HALVES;      // tiles per block
    static constexpr int SLOTS = TB + 1;        // + one halo slot
    static constexpr int PIX = 8 * SLOTS;       // (4 tile rows) x (even, odd) x slots
    static constexpr int ITEMS = PIX * 4;       // 16-byte pieces of a stage
    static constexpr int DMA = (ITEMS + 63) / 64;   // wave-wide LDS-DMA instructions per stage (1 KB each; the last one is partly used)
    static constexpr int STAGE = DMA * 256;     // floats per stage
    static constexpr int ROW = 2 * SLOTS * W_CK;    // floats per tile row i
};
template <int NH, int HALVES>
constexpr int wino_lds_floats() {
    const int loop = W_ZERO + 2 * WinoGeom<HALVES>::STAGE, epi = 4 * 2 * WinoGeom<HALVES>::TB * (32 * NH + 4);
    return loop > epi ? loop : epi;
}

typedef __attribute__((address_space(3))) void wlds_void;

// (body in a __device__ function: with the DMA builtin inside the __global__ template hipcc 7.2 drops the kernel's host stub)
template <int NH, int HALVES, bool BNB>
__device__ __forceinline__ void conv3x3_winograd_body(const WinoParams& p, float* smem) {
    using G = WinoGeom<HALVES>;
    constexpr int W_TB = G::TB, W_SLOTS = G::SLOTS, W_ITEMS = G::ITEMS, W_DMA = G::DMA, W_STAGE = G::STAGE, W_ROW = G::ROW;
    constexpr int NT = 256 * NH, BN = 32 * NH, NW = 4 * NH;
    constexpr int NLD = (W_DMA + NW - 1) / NW;             // DMA instructions per wave per stage
    float* Rs = smem + W_ZERO;                             // [2][4 rows][2 parities][65 slots][16 channels], chunk-swizzled

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave / NH, nh = wave % NH;

    // XCD-aware tile order: block b runs on XCD b % 8; each XCD gets a contiguous run of tiles with the m-tile fastest, so the
    // blocks of an XCD share one filter slice (16 * Cin * BN * 4 bytes <= 2 MB) in their L2
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q8 = nblk >> 3, r8 = nblk & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int n_tile = t / p.m_tiles, m_tile = t - n_tile * p.m_tiles;
    const int m0 = m_tile * W_TB, n0 = n_tile * BN;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

    // ---- staging by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write pass).  The destination of a wave instruction is
    // lane-linear (base + lane * 16 bytes), so the chunk swizzle is applied on the SOURCE side: LDS position q = (pixel q >> 2, chunk
    // position q & 3) receives the pixel's global chunk (q & 3) ^ ((slot >> 2) & 3).  pixel = (row i, parity, slot).  Outside the image
    // / launch the offset is out of range and the DMA writes zeros.
    unsigned goff[NLD];                                    // byte offset of the lane's piece in the first stage (WOOB: zeros)
#pragma unroll
    for (int u = 0; u < NLD; ++u) {
        const int q = (wave + NW * u) * 64 + lane;
        goff[u] = WOOB;
        if (q < W_ITEMS) {
            const int cpos = q & 3, pix = q >> 2;
            const int ipar = pix / W_SLOTS, slot = pix - ipar * W_SLOTS;
            const int i = ipar >> 1, par = ipar & 1;
            const int chunk = cpos ^ ((slot >> 2) & 3);
            const int m = m0 + slot - par;                 // even array: slot s = tile m0 + s;  odd array: slot s = tile m0 + s - 1
            if (m >= 0 && m < p.Mtiles) {
                const int b = m / p.tpi, r = m - b * p.tpi;
                const int ty = r / p.TW, tx = r - ty * p.TW;
                const int yy = 2 * ty - 1 + i, xx = 2 * tx + par;
                if ((unsigned)yy < (unsigned)p.H && xx < p.W) goff[u] = (unsigned)(((b * p.H + yy) * p.W + xx) * p.Cin + chunk * 4) << 2;
            }
        }
    }
    auto stage_dma = [&](int buf, int st) {
        const bool live = st < p.stages;
#pragma unroll
        for (int u = 0; u < NLD; ++u)
            if (wave + NW * u < W_DMA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (wlds_void*)(Rs + buf * W_STAGE + (wave + NW * u) * 256), 16,
                                                         (live && goff[u] != WOOB) ? goff[u] + (unsigned)st * (W_CK * 4) : WOOB, 0, 0, 0);
    };

    // ---- fragment addressing: lane = (tile l & 31 of a half, channel quad l >> 5) ---------------------------------------------------
    // ra[half][j]: float index (relative to Rs, tile row 0, stage 0, first 8-channel step) of column j of the lane's tile; the second
    // step of a stage is the same index ^ 8.  A border column points at the zero pixel (index -16).
    const int h = lane >> 5;
    int ra[HALVES][4];
#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
        const int tl = 32 * half + (lane & 31);
        const int m = m0 + tl;
        const int mm = m < p.Mtiles ? m : p.Mtiles - 1;
        const int r = mm % p.tpi;
        const int tx = r % p.TW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int par = (j & 1) ^ 1;                   // j = 0, 2: odd pixels;  j = 1, 3: even pixels
            const int slot = tl + (j >> 1);                // j = 0: odd[t], 1: even[t], 2: odd[t + 1], 3: even[t + 1]
            const bool border = (j == 0 && tx == 0) || (j == 3 && tx == p.TW - 1);
            ra[half][j] = border ? -W_ZERO : (par * W_SLOTS + slot) * W_CK + ((h ^ ((slot >> 2) & 3)) << 2);
        }
    }
    // row transform of this wave: t = d[ia] + sg * d[ib]
    const int ia = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int ib = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float sgn = xi == 1 ? 1.f : -1.f;
    const int roa = ia * W_ROW, rob = ib * W_ROW;

#ifdef VATL_ABLATION
    const int abl = p.ablate;
#else
    constexpr int abl = 0;
#endif
    // ---- U fragments: [n_tile][step][position][nh][lane][4] ----------------------------------------------------------------------------
    const int steps = p.stages * 2;
    // (the packing groups 32 p.nhp channels per filter tile; this block's 32-channel group nh_g of it)
    const int n32 = n_tile * NH + nh, ut = n32 / p.nhp, nh_g = n32 - ut * p.nhp;
    const unsigned ubase = (unsigned)((((ut * steps) * 16 + 4 * xi) * p.nhp + nh_g) * 64 + lane) << 4;   // bytes; + step * 16*nhp*1024 + nu * nhp*1024
    const unsigned ustep = 16u * p.nhp * 1024u, unu = p.nhp * 1024u;
    auto u_load = [&](f32x4 (&dst)[4], int step) {
        const bool live = step < steps && !(abl & 4);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) dst[nu] = wbuf_load4(ur, live ? ubase + (unsigned)step * ustep + nu * unu : WOOB);
    };

    f32x16 acc[4][HALVES];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu)
#pragma unroll
        for (int half = 0; half < HALVES; ++half)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nu][half][e] = 0.f;

    if (tid < W_ZERO / 4) *reinterpret_cast<f32x4*>(&smem[tid * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ua[4], ub[4];
    stage_dma(0, 0);
    u_load(ua, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's DMA pieces have landed
    __syncthreads();

    auto step_mfma = [&](const float* Rb, int x8, const f32x4 (&uu)[4]) {
#pragma unroll
        for (int half = 0; half < HALVES; ++half) {
            f32x4 tc[4];
            if (abl & 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) tc[j] = f32x4{1.f + j, 2.f, 3.f, 4.f + x8};
            } else
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool z = ra[half][j] < 0;            // the zero pixel has no rows / stages
                const float* pa = z ? smem : Rb + roa + (ra[half][j] ^ x8);
                const float* pb = z ? smem : Rb + rob + (ra[half][j] ^ x8);
                const f32x4 da = *reinterpret_cast<const f32x4*>(pa);
                const f32x4 db = *reinterpret_cast<const f32x4*>(pb);
                tc[j] = da + sgn * db;
            }
            f32x4 v[4];
            v[0] = tc[0] - tc[2]; v[1] = tc[1] + tc[2]; v[2] = tc[2] - tc[1]; v[3] = tc[1] - tc[3];
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int nu = 0; nu < 4; ++nu)
                    acc[nu][half] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu][tt], uu[nu][tt], acc[nu][half], 0, 0, 0);
        }
    };

    for (int st = 0; st < p.stages; ++st) {
        const int buf = st & 1;
        const float* Rb = Rs + buf * W_STAGE;
        // the scheduler barriers keep the requests where they are written: without them hipcc sinks the staging loads to just
        // before their LDS writes and the filter loads to just before their first MFMA (latency fully exposed)
        if (st + 1 < p.stages && !(abl & 8)) stage_dma(buf ^ 1, st + 1);
        u_load(ub, 2 * st + 1);
        __builtin_amdgcn_sched_barrier(0);
        step_mfma(Rb, 0, ua);
        __builtin_amdgcn_sched_barrier(0);
        u_load(ua, 2 * st + 2);
        __builtin_amdgcn_sched_barrier(0);
        step_mfma(Rb, 8, ub);
        __builtin_amdgcn_sched_barrier(0);
        if (!(abl & 16)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of the next stage have landed
            __syncthreads();
        }
    }

#ifdef VATL_ABLATION
    if (abl & 16) __syncthreads();
    if (abl & 1) {                     // keep the accumulators alive, skip the write-out
        float sacc = 0.f;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int half = 0; half < HALVES; ++half)
#pragma unroll
                for (int e = 0; e < 16; ++e) sacc += acc[nu][half][e];
        if (sacc == 12345.678f) p.y[0] = sacc;
        return;
    }
#endif
    // ---- output transform -----------------------------------------------------------------------------------------------------------------
    // nu sum in registers: P[b = 0] = M0 + M1 + M2, P[b = 1] = M1 - M2 - M3;  Ps[xi][b][tile][n] in LDS
    constexpr int LDP = BN + 4;
    float* Ps = smem;
#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
        const int cl = 32 * nh + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = 32 * half + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const float m0v = acc[0][half][e], m1v = acc[1][half][e], m2v = acc[2][half][e], m3v = acc[3][half][e];
            Ps[((xi * 2 + 0) * W_TB + row) * LDP + cl] = m0v + m1v + m2v;
            Ps[((xi * 2 + 1) * W_TB + row) * LDP + cl] = m1v - m2v - m3v;
        }
    }
    __syncthreads();

    // xi sum: Y[a = 0] = P0 + P1 + P2, Y[a = 1] = P1 - P2 - P3 per (tile, b, channel quad); full-row 16-byte stores
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.res ? p.y_bytes : 0u, 0x00020000);
    constexpr int C4 = BN / 4;                             // channel quads per tile row
    constexpr int TPP = NT / (2 * C4);                     // tiles per pass (16)
    constexpr int NP = W_TB / TPP;                         // passes
    const int c4 = tid % C4, bq = (tid / C4) & 1, tl0 = tid / (2 * C4);
    const int n = n0 + c4 * 4;
    const bool nv = n < p.Cout;
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, nul = {0.f, 0.f, 0.f, 0.f};
    const f32x4 sc = (nv && p.scale) ? *reinterpret_cast<const f32x4*>(p.scale + n) : one;
    const f32x4 bi = (nv && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + n) : nul;
    const float lo = p.relu ? 0.f : -INFINITY;
    f32x4 ssum = nul, ssq = nul;
    const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNB ? p.bz : p.x), 0, BNB ? p.y_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNB ? p.bmy : p.x), 0, (BNB && p.bmy) ? p.y_bytes : 0u, 0x00020000);
    f32x4 mu = nul, is = nul, msc = nul, mbi = one;       // no mask: 0 * z + 1 > 0
    if (BNB && nv) {
        mu = *reinterpret_cast<const f32x4*>(p.bmu + n); is = *reinterpret_cast<const f32x4*>(p.bis + n);
        if (p.bsc) { msc = *reinterpret_cast<const f32x4*>(p.bsc + n); mbi = *reinterpret_cast<const f32x4*>(p.bbi + n); }
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        const int tl = tl0 + u * TPP;
        const int m = m0 + tl;
        unsigned off[2] = {WOOB, WOOB};
        if (nv && m < p.Mtiles) {
            const int b = m / p.tpi, r = m - b * p.tpi;
            const int ty = r / p.TW, tx = r - ty * p.TW;
            const int xx = 2 * tx + bq;
            if (xx < p.W) {
                const int pix = (b * p.H + 2 * ty) * p.W + xx;
                off[0] = (unsigned)(pix * p.Cout + n) << 2;
                if (2 * ty + 1 < p.H) off[1] = (unsigned)((pix + p.W) * p.Cout + n) << 2;
            }
        }
        f32x4 rs[2] = {nul, nul};
        if (p.res) { rs[0] = wbuf_load4(rr, off[0]); rs[1] = wbuf_load4(rr, off[1]); }
        f32x4 pq[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) pq[k] = *reinterpret_cast<const f32x4*>(&Ps[((k * 2 + bq) * W_TB + tl) * LDP + c4 * 4]);
        f32x4 yv[2];
        yv[0] = pq[0] + pq[1] + pq[2];
        yv[1] = pq[1] - pq[2] - pq[3];
        if constexpr (BNB) {
            f32x4 zt[2], yt[2] = {nul, nul};
            zt[0] = wbuf_load4(zr, off[0]); zt[1] = wbuf_load4(zr, off[1]);
            if (p.bmy) { yt[0] = wbuf_load4(mr, off[0]); yt[1] = wbuf_load4(mr, off[1]); }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                f32x4 gq;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float d = yv[a][c] + rs[a][c];
                    const bool on = p.bmy ? yt[a][c] > 0.f : fmaf(zt[a][c], msc[c], mbi[c]) > 0.f;
                    gq[c] = (on && off[a] != WOOB) ? d : 0.f;
                    ssum[c] += gq[c];
                    ssq[c] += gq[c] * ((zt[a][c] - mu[c]) * is[c]);
                }
                wbuf_store4(yr, off[a], gq);
            }
        } else {
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = fmaxf(yv[a][c] * sc[c] + bi[c] + rs[a][c], lo);
            wbuf_store4(yr, off[a], o);
            if (p.stats && off[a] != WOOB) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { ssum[c] += o[c]; ssq[c] += o[c] * o[c]; }
            }
        }
        }
    }
    if (p.stats) {
        // BatchNorm batch statistics of the pixels just stored: one (sum, sum^2) double pair per (m-tile, channel), fixed order
        __syncthreads();
        f32x4* sh = reinterpret_cast<f32x4*>(smem);
        sh[tid] = ssum; sh[NT + tid] = ssq;
        __syncthreads();
        if (tid < C4) {
            double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
#pragma unroll 2
            for (int k = 0; k < NT / C4; ++k) {
                const f32x4 a = sh[k * C4 + tid], b = sh[NT + k * C4 + tid];
#pragma unroll
                for (int c = 0; c < 4; ++c) { ds[c] += a[c]; dq[c] += b[c]; }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int nn = n0 + tid * 4 + c;
                if (nn < p.Cout) {
                    p.stats[((long long)m_tile * p.Cout + nn) * 2 + 0] = ds[c];
                    p.stats[((long long)m_tile * p.Cout + nn) * 2 + 1] = dq[c];
                }
            }
        }
    }
}

// <NH = 2, HALVES = 2>: 64 tiles x 64 channels, 8 waves, one block per CU (large launches);  <1, 2>: 64 tiles x 32 channels (Cout <= 32),
// two blocks per CU;  <1, 1>: 32 tiles x 32 channels, 64 accumulator registers, three blocks per CU — small launches (fine-tune batches),
// where the large tile leaves CUs idle.  The arithmetic of a tile does not depend on the configuration: same bits.
template <int NH, int HALVES, bool BNB>
__global__ __launch_bounds__(256 * NH, NH == 1 ? (HALVES == 1 ? 3 : 2) : 1) void conv3x3_winograd_kernel(WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv3x3_winograd_body<NH, HALVES, BNB>(p, smem);
}

// U = G g G^T (winograd_pack.h): one block per (32 output channels, 8 input channels)
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int NH, int mode, int w_i) {
    wino_pack_block(w, out, mode, w_i, Cout, Cin, NH, blockIdx.x, threadIdx.x);
}

static std::atomic<unsigned> g_wino_lds_done[8];
static std::atomic<int> g_wino_cfg{0};            // vatl_tune_set(18, v): 0 / 2 = 32 x 32 blocks (default), 1 = 64-tile blocks
int wino_set_cfg(int v) { g_wino_cfg.store(v, std::memory_order_relaxed); return 0; }
static std::atomic<int> g_wino_ablate{0};
int wino_set_ablate(int bits) { g_wino_ablate.store(bits, std::memory_order_relaxed); return 0; }

template <int NH, int HALVES, bool BNB>
static int launch_wino(WinoParams& p, int64_t* row_blocks_used, hipStream_t st) {
    constexpr int smem = wino_lds_floats<NH, HALVES>() * (int)sizeof(float);
    auto kern = conv3x3_winograd_kernel<NH, HALVES, BNB>;
    if (int rc = ensure_dynamic_lds((const void*)kern, smem, g_wino_lds_done[(NH - 1) * 2 + (HALVES - 1) + (BNB ? 4 : 0)], "conv3x3_winograd")) return rc;
    p.m_tiles = cdiv(p.Mtiles, 32 * HALVES); p.n_tiles = cdiv(p.Cout, 32 * NH);
    if (row_blocks_used) *row_blocks_used = p.m_tiles;
    hipLaunchKernelGGL(kern, dim3(p.m_tiles * p.n_tiles), dim3(256 * NH), smem, st, p);
    return check_launch("conv3x3_winograd");
}

}  // namespace vatl

using namespace vatl;

static int wino_nh(int Cout) { return Cout <= 32 ? 1 : 2; }

extern "C" int vatl_winograd_cout_pad(int Cout) {
    const int bn = 32 * wino_nh(Cout);
    return (Cout + bn - 1) / bn * bn;
}

extern "C" int64_t vatl_winograd_weight_floats(int Cout, int Cin) { return (int64_t)vatl_winograd_cout_pad(Cout) * Cin * 16; }

extern "C" int vatl_pack_winograd_weight(const float* w, float* u, int Cout, int Cin, int data_gradient, void* stream) {
    if (!w || !u || Cout <= 0 || Cin <= 0) return fail(VATL_EINVAL, "pack_winograd_weight: null pointer or empty filter");
    if (Cin % 16 != 0) return fail(VATL_EINVAL, "pack_winograd_weight: Cin %d must be a multiple of 16", Cin);
    const int pad = vatl_winograd_cout_pad(Cout);
    // data_gradient: the caller passes the FORWARD filter w[O][I][3][3] and asks for the filter of dX = conv(dY, rot180(w)^T):
    // Cout = I (channels of dX), Cin = O (channels of dY)
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)(pad / 32 * (Cin / 8))), dim3(256), 0, (hipStream_t)stream, w, u, Cout, Cin, wino_nh(Cout),
                       data_gradient ? 1 : 0, data_gradient ? Cout : Cin);
    return check_launch("wino_pack");
}

struct WinoBn { const float *z, *mask_y, *scale, *bias, *mean, *invstd; };

static int winograd_impl(const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y, double* stats,
                         int64_t* row_blocks_used, int N, int H, int W, int Cin, int Cout, int relu, void* stream, const WinoBn* fuse = nullptr) {
    if (!x || !u || !y || N <= 0 || H <= 0 || W <= 0) return fail(VATL_EINVAL, "conv3x3_winograd: null pointer or empty batch");
    if (Cin % 16 != 0 || (Cout & 3)) return fail(VATL_EINVAL, "conv3x3_winograd: Cin %d must be a multiple of 16 and Cout %d of 4", Cin, Cout);
    WinoParams p{};
    p.x = x; p.u = u; p.scale = scale; p.bias = bias; p.res = residual; p.y = y; p.stats = stats;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
    p.TH = (H + 1) / 2; p.TW = (W + 1) / 2; p.tpi = p.TH * p.TW;
    const long long mt = (long long)N * p.tpi, xe = (long long)N * H * W * Cin, ye = (long long)N * H * W * Cout;
    const long long ue = vatl_winograd_weight_floats(Cout, Cin);
    if (xe >= (1LL << 30) || ye >= (1LL << 30) || ue >= (1LL << 30) || mt >= (1LL << 30))
        return fail(VATL_EINVAL, "conv3x3_winograd: a tensor exceeds 2^30 elements (32-bit buffer offsets); split the batch");
    p.Mtiles = (int)mt;
    p.nhp = wino_nh(Cout);
    p.stages = Cin / W_CK;
    p.x_bytes = (unsigned)(xe * 4); p.y_bytes = (unsigned)(ye * 4); p.u_bytes = (unsigned)(ue * 4);
    p.ablate = g_wino_ablate.load(std::memory_order_relaxed);
    if (fuse) { p.bz = fuse->z; p.bmy = fuse->mask_y; p.bsc = fuse->scale; p.bbi = fuse->bias; p.bmu = fuse->mean; p.bis = fuse->invstd; }
    // The 32 x 32 configuration (three blocks per CU, 12 waves) beats the 64-tile ones at every size measured on MI355X — 1024 crops:
    // l1.c2 1245 vs 1333 us, l4.c2 967 vs 1029, hr.b32 476 vs 537;  120 crops: l3.c2 150 vs 197, l4.c2 142 vs 191 (tools/wino_bench.py
    // --cfg 2 / 1): the blocks of a CU overlap each other's prologue, barriers and output transform.  The large ones stay selectable.
    const int cfg = g_wino_cfg.load(std::memory_order_relaxed);
    const bool small = cfg != 1;
    hipStream_t st = (hipStream_t)stream;
    if (small) return fuse ? launch_wino<1, 1, true>(p, row_blocks_used, st) : launch_wino<1, 1, false>(p, row_blocks_used, st);
    if (p.nhp == 1) return fuse ? launch_wino<1, 2, true>(p, row_blocks_used, st) : launch_wino<1, 2, false>(p, row_blocks_used, st);
    return fuse ? launch_wino<2, 2, true>(p, row_blocks_used, st) : launch_wino<2, 2, false>(p, row_blocks_used, st);
}

extern "C" int vatl_conv3x3_winograd_fwd(const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y,
                                         int N, int H, int W, int Cin, int Cout, int relu, void* stream) {
    return winograd_impl(x, u, scale, bias, residual, y, nullptr, nullptr, N, H, W, Cin, Cout, relu, stream);
}

extern "C" int64_t vatl_winograd_stats_row_blocks(int64_t N, int H, int W) { return (N * ((H + 1) / 2) * ((W + 1) / 2) + 31) / 32; }   // capacity

extern "C" int vatl_conv3x3_winograd_fwd_stats(const float* x, const float* u, float* y, double* stats, int64_t* row_blocks_used, int N, int H,
                                               int W, int Cin, int Cout, void* stream) {
    if (!stats || !row_blocks_used) return fail(VATL_EINVAL, "conv3x3_winograd_fwd_stats: null statistics buffer");
    return winograd_impl(x, u, nullptr, nullptr, nullptr, y, stats, row_blocks_used, N, H, W, Cin, Cout, 0, stream);
}

// Data-gradient launch fused with the reduction pass of the consumer layer's BatchNorm backward: the Winograd counterpart of
// vatl_conv2d_fwd_ex_bnbwd (same masks, same statistics layout; u = the data-gradient packing of the filter).
extern "C" int vatl_conv3x3_winograd_fwd_bnbwd(const float* x, const float* u, const float* residual, float* y, int N, int H, int W, int Cin, int Cout,
                                               const float* bn_z, const float* bn_mask_y, const float* bn_scale, const float* bn_bias,
                                               const float* bn_mean, const float* bn_invstd, double* stats, int64_t* row_blocks_used, void* stream) {
    if (!bn_z || !stats || !row_blocks_used || !bn_mean || !bn_invstd || (bn_scale && !bn_bias))
        return fail(VATL_EINVAL, "conv3x3_winograd_fwd_bnbwd: needs z, mean, invstd and a statistics buffer");
    const WinoBn bn{bn_z, bn_mask_y, bn_scale, bn_bias, bn_mean, bn_invstd};
    return winograd_impl(x, u, nullptr, nullptr, residual, y, stats, row_blocks_used, N, H, W, Cin, Cout, 0, stream, &bn);
}
