// Weight gradients on the Winograd route (the backward counterpart of conv_winograd.hip):
//   * 3x3 / stride 1 / pad 1 conv (MO = 2):        dW = G^T [ sum_tiles (A dY A^T) .* (B^T d B) ] G
//   * ConvTranspose2d(4, 2, 1), per phase (MO = 3): the same with the 2x2 phase filter, dY = that phase's 3x3 output-gradient tile
// i.e. per transform position p = (xi, nu) one GEMM  dU_p[n][c] = sum_t Z_p[t][n] V_p[t][c]  with the reduction over ALL tiles of the
// batch: 16 multiplies per (tile, n, c) where the direct sum has 36 (conv_wgrad.hip).  Exact fp32 products, fp32 accumulation.
//
// Block = 32 output channels (n) x 32 input channels (c) x 16 positions x one contiguous range of tiles (split); 4 waves, wave xi owns
// the positions (xi, 0..3) = 64 accumulator registers.  The MFMA reduces over tile pairs: A operand Z_p[tile][n], B operand V_p[tile][c].
// Per stage of TK tiles the raw pixels are staged by LDS-DMA, channel-fastest: the input x as in the forward kernel (four tile rows,
// one slot per distinct pixel column), the output gradient as MO rows x (TK MO) pixels; pixels outside the image / the split are
// out-of-range DMA offsets = zeros.  A lane (channel l & 31, tile parity l >> 5) reads its channel of 8 input pixels and <= MO^2 gradient
// pixels per tile pair (ds_read_b32: 32 consecutive channels = 128 contiguous bytes), forms V (row transform by xi, then the four nu)
// and Z in registers, and issues four MFMAs.
// Measured and dropped: a 32 x 64 block of 8 waves (same 16 waves per CU, half the staging-address work and workspace traffic per MFMA):
// equal to 10 % slower on the 3x3 layers, 15 - 30 % slower on the transposed convs; division-free staging addresses for maps at least TK - 1
// tiles wide (row bases in scalar registers, one compare per item): 2 - 5 % slower at 128 VGPRs.  (An ablation with constant staging
// addresses runs 17 % faster — but that is the memory system seeing the same few lines, not the address arithmetic.)
// The nu side of G^T . G is applied to the accumulators in registers (4 tiles -> 3, or 2 for the phase filters); each block writes its
// partial sums into its slice of a workspace and winograd_wgrad_finish_kernel adds the slices in split order, applies the xi side and
// writes the filter in the tensor's own layout — no atomics, bitwise reproducible.
#include "common.h"

#include <algorithm>
#include <atomic>

namespace vatl {

struct WinoWgradParams {
    const float* x;                   // (N, H, W, Cx): the conv input (transposed conv: its input)
    const float* g;                   // (N, OH, OW, Cn): gradient of the conv output
    float* part;                      // [split][phase][set][xi][CnPad][CxPad]
    int N, H, W, Cx, Cn;
    int CnPad, CxPad;
    int TH, TW, tpi, Mtiles;
    int n_tiles, c_tiles, splits, tps; // tiles per split (a multiple of TK)
    int RW, ns;                       // slot geometry of the x stage (conv_winograd.hip)
    int ndma_x, ndma_g, stage_floats, g_floats_off;   // LDS-DMA instructions per stage, floats per stage, offset of the gradient pixels
    int OH, OW, os;                   // gradient pixel of grid point (y, x) = (y os + phase_y, x os + phase_x)
    int deconv;
    float inv_TW, inv_TH, inv_RW, inv_ns;
    unsigned x_bytes, g_bytes;
    // staging addresses of every stage of the launch, written by winograd_wgrad_table_kernel right before it (null: computed in the kernel):
    // tab_x [x phase][stage][ndma_x * 64] byte offsets for channel tile 0, tab_g [stage][ndma_g * 64] for channel tile 0 / phase (0, 0); 0xFFFFFFFF = zeros
    const unsigned* tab_x;
    const unsigned* tab_g;
    int stages_total;
};

constexpr unsigned WWOOB = 0xFFFFFFFFu;
typedef __attribute__((address_space(3))) void wwlds_void;

// The staging addresses of a launch depend on its geometry only — and forming them in the main kernel (two divisions and ~25 vector instructions per 16-byte
// request, ~240 per stage and lane) costs 19 - 26 % of its time: vector instructions are not hidden behind the MFMAs on this hardware (profiles/r04_notes.md;
// with the arithmetic removed: deconv3 1648 -> 1217 us, r152.l3.c2 426 -> 347, with the arithmetic kept and the same stale addresses used: unchanged).  So a
// small kernel writes them once per launch into the workspace and the blocks of all channel tiles and splits read them back (one 4-byte load per request).
template <int MO, int TK, int NBN>
__global__ __launch_bounds__(256) void winograd_wgrad_table_kernel(WinoWgradParams p, unsigned* tab_x, unsigned* tab_g) {
    constexpr int GP = TK * MO;
    const int sidx = blockIdx.x, xphase = blockIdx.y;
    const int t0 = sidx * TK;
    const int py = xphase >> 1, px = xphase & 1;
    const int pad_y = p.deconv ? 1 - py : 1, pad_x = p.deconv ? 1 - px : 1;
    const int gr0 = t0 / p.TW, pos0 = MO * (t0 - gr0 * p.TW);
    const int nx = p.ndma_x * 64, ng = p.ndma_g * 64;
    for (int q = threadIdx.x; q < nx; q += 256) {
        const int pix = q >> 3, i = pix / p.ns, slot = pix - i * p.ns;
        unsigned off = WWOOB;
        if (i < 4) {
            const int P = slot + pos0, rr = P / p.RW, pos = P - rr * p.RW;
            const int grow = gr0 + rr, b = grow / p.TH, ty = grow - b * p.TH;
            const int yy = MO * ty - pad_y + i, xx = pos - pad_x;
            if (b < p.N && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) off = (unsigned)(((b * p.H + yy) * p.W + xx) * p.Cx + (q & 7) * 4) << 2;
        }
        tab_x[((long long)xphase * p.stages_total + sidx) * nx + q] = off;
    }
    if (xphase == 0) {
        for (int q = threadIdx.x; q < ng; q += 256) {
            const int pix = q / (8 * NBN), a = pix / GP, sl = pix - a * GP, tl = sl / MO, bcol = sl - tl * MO;
            const int t = t0 + tl;
            unsigned off = WWOOB;
            if (a < MO && t < p.Mtiles) {
                const int grow = t / p.TW, tx = t - grow * p.TW, b = grow / p.TH, ty = grow - b * p.TH;
                const int yy = MO * ty + a, xx = MO * tx + bcol;
                if (yy < p.H && xx < p.W) off = (unsigned)(((b * p.OH + yy * p.os) * p.OW + xx * p.os) * p.Cn + (q % (8 * NBN)) * 4) << 2;
            }
            tab_g[(long long)sidx * ng + q] = off;
        }
    }
}

// NBN = 2: 64 output channels per block (two 32-channel halves, 128 accumulator registers, two blocks per CU).  The kernel is bound by
// vector-instruction issue, not by the matrix pipe (36 % busy in round 3: ~9 vector instructions per MFMA — tile addressing, the 8 + 4 LDS
// reads and both transforms of every tile pair); with two halves the input side V = B^T d B of a tile pair — the expensive one — and the
// tile addressing are formed once for twice the MFMAs, only the cheap gradient side Z = A dY A^T is formed per half.
template <int MO, int TK, int NBN>
__device__ __forceinline__ void winograd_wgrad_body(const WinoWgradParams& p, float* smem) {
    constexpr int NS = MO == 2 ? 3 : 2;                    // accumulator sets after the nu side of G
    constexpr int GP = TK * MO;                            // gradient pixels per stage row
    constexpr int GC = 32 * NBN;                           // gradient channels staged per pixel
    constexpr int NLX = 6, NLG = (MO * GP * 8 * NBN + 255) / 256;   // DMA instructions per wave per stage, at most (x: ns <= 48)
    const int tid = threadIdx.x, lane = tid & 63, xi = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (n tile, c tile, phase, split), the channel tiles fastest: the blocks of one split read the same pixels
    int bid = blockIdx.x;
    const int n_tile = bid % p.n_tiles; bid /= p.n_tiles;
    const int c_tile = bid % p.c_tiles; bid /= p.c_tiles;
    const int phases = p.deconv ? 4 : 1;
    const int phase = bid % phases;
    const int split = bid / phases;
    const int n0 = n_tile * GC, c0 = c_tile * 32;
    const int py = phase >> 1, px = phase & 1;
    const int pad_y = p.deconv ? 1 - py : 1, pad_x = p.deconv ? 1 - px : 1;
    const int ooy = p.deconv ? py : 0, oox = p.deconv ? px : 0;
    const int t_begin = split * p.tps, t_end = min(t_begin + p.tps, p.Mtiles);

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g), 0, p.g_bytes, 0x00020000);
    // table mode: the tables hold the offsets of channel tile 0 (and gradient phase (0, 0)); this block's tile / phase is in the descriptors' base addresses
    const bool tabled = p.tab_x != nullptr;
    const unsigned xsh = (unsigned)c0 * 4u, gsh = (unsigned)((ooy * p.OW + oox) * p.Cn + n0) * 4u;
    const __amdgpu_buffer_rsrc_t xrt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x) + c0, 0, p.x_bytes - xsh, 0x00020000);
    const __amdgpu_buffer_rsrc_t grt = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.g) + (gsh >> 2), 0, p.g_bytes - gsh, 0x00020000);
    const unsigned* tx_base = tabled ? p.tab_x + ((long long)(p.deconv ? phase : 0) * p.stages_total) * (p.ndma_x * 64) + xi * 64 + lane : nullptr;
    const unsigned* tg_base = tabled ? p.tab_g + xi * 64 + lane : nullptr;
    unsigned offx[NLX], offg[NLG];                          // the NEXT stage_dma's addresses, loaded a stage ahead
    auto load_offsets = [&](int t0) {
        const int sidx = t0 / TK;                           // (t0 is a multiple of TK: splits start on stage boundaries)
#pragma unroll
        for (int u = 0; u < NLX; ++u) offx[u] = (xi + 4 * u < p.ndma_x) ? tx_base[(long long)sidx * (p.ndma_x * 64) + u * 256] : WWOOB;
#pragma unroll
        for (int u = 0; u < NLG; ++u) offg[u] = (xi + 4 * u < p.ndma_g) ? tg_base[(long long)sidx * (p.ndma_g * 64) + u * 256] : WWOOB;
    };

    // ---- staging items of this lane (constant over the stages): x item = (row i, slot, 16-byte chunk), gradient item = (row a, tile, col b, chunk)
    int xi_i[NLX], xi_slot[NLX], xi_c[NLX];
#pragma unroll
    for (int u = 0; u < NLX; ++u) {
        const int q = (xi + 4 * u) * 64 + lane;
        const int pix = q >> 3;
        const int i = fast_div(pix, p.inv_ns);
        xi_i[u] = i; xi_slot[u] = pix - i * p.ns; xi_c[u] = c0 + (q & 7) * 4;
    }
    int gi_a[NLG], gi_t[NLG], gi_b[NLG], gi_c[NLG];
#pragma unroll
    for (int u = 0; u < NLG; ++u) {
        const int q = (xi + 4 * u) * 64 + lane;
        const int pix = q / (8 * NBN);
        const int a = pix / GP, sl = pix - a * GP;
        gi_a[u] = a; gi_t[u] = sl / MO; gi_b[u] = sl - gi_t[u] * MO; gi_c[u] = n0 + (q % (8 * NBN)) * 4;
    }
    auto stage_dma = [&](int buf, int t0) {
        float* Xs = smem + buf * p.stage_floats;
        float* Gs = Xs + p.g_floats_off;
        if (tabled) {
#pragma unroll
            for (int u = 0; u < NLX; ++u)
                if (xi + 4 * u < p.ndma_x) __builtin_amdgcn_raw_ptr_buffer_load_lds(xrt, (wwlds_void*)(Xs + (xi + 4 * u) * 256), 16, offx[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NLG; ++u)
                if (xi + 4 * u < p.ndma_g) __builtin_amdgcn_raw_ptr_buffer_load_lds(grt, (wwlds_void*)(Gs + (xi + 4 * u) * 256), 16, offg[u], 0, 0, 0);
            return;
        }
        const int gr0 = fast_div(t0, p.inv_TW);            // global tile row (image, ty) of the stage's first tile
        const int pos0 = MO * (t0 - gr0 * p.TW);
#pragma unroll
        for (int u = 0; u < NLX; ++u) {
            if (xi + 4 * u < p.ndma_x) {
                unsigned off = WWOOB;
                if (xi_i[u] < 4 && xi_c[u] < p.Cx) {
                    const int P = xi_slot[u] + pos0;
                    const int rr = fast_div(P, p.inv_RW), pos = P - rr * p.RW;
                    const int grow = gr0 + rr;
                    const int b = fast_div(grow, p.inv_TH), ty = grow - b * p.TH;
                    const int yy = MO * ty - pad_y + xi_i[u], xx = pos - pad_x;
                    if (b < p.N && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
                        off = (unsigned)(((b * p.H + yy) * p.W + xx) * p.Cx + xi_c[u]) << 2;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (wwlds_void*)(Xs + (xi + 4 * u) * 256), 16, off, 0, 0, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < NLG; ++u) {
            if (xi + 4 * u < p.ndma_g) {
                unsigned off = WWOOB;
                const int t = t0 + gi_t[u];
                if (gi_a[u] < MO && t < t_end && gi_c[u] < p.Cn) {
                    const int grow = fast_div(t, p.inv_TW), tx = t - grow * p.TW;
                    const int b = fast_div(grow, p.inv_TH), ty = grow - b * p.TH;
                    const int yy = MO * ty + gi_a[u], xx = MO * tx + gi_b[u];
                    if (yy < p.H && xx < p.W)
                        off = (unsigned)(((b * p.OH + yy * p.os + ooy) * p.OW + xx * p.os + oox) * p.Cn + gi_c[u]) << 2;
                }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(gr, (wwlds_void*)(Gs + (xi + 4 * u) * 256), 16, off, 0, 0, 0);
            }
        }
    };

    // row transforms of this wave.  V: t = d[ia] + vs * d[ib] (B^T rows d0 - d2, d1 + d2, d2 - d1, d1 - d3);
    // Z: rows of A (MO = 2: [1 0], [1 1], [1 -1], [0 -1];  MO = 3: [1 0 0], [1 1 1], [1 -1 1], [0 0 1])
    const int ia = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int ib = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float vs = xi == 1 ? 1.f : -1.f;
    float za[3];
    if (MO == 2) { za[0] = xi == 3 ? 0.f : 1.f; za[1] = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f); za[2] = 0.f; }
    else { za[0] = xi == 3 ? 0.f : 1.f; za[1] = xi == 1 ? 1.f : (xi == 2 ? -1.f : 0.f); za[2] = xi == 0 ? 0.f : 1.f; }
    const int ch = lane & 31, kh = lane >> 5;
    const int roa = ia * p.ns * 32, rob = ib * p.ns * 32;

    f32x16 acc[NBN][4];
#pragma unroll
    for (int hh = 0; hh < NBN; ++hh)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[hh][nu][e] = 0.f;

    if (t_begin < t_end) {
        if (tabled) load_offsets(t_begin);
        stage_dma(0, t_begin);
        if (tabled && t_begin + TK < t_end) load_offsets(t_begin + TK);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    int buf = 0;
    for (int t0 = t_begin; t0 < t_end; t0 += TK, buf ^= 1) {
        if (t0 + TK < t_end) {
            stage_dma(buf ^ 1, t0 + TK);
            if (tabled && t0 + 2 * TK < t_end) load_offsets(t0 + 2 * TK);        // (arrive with the stage: the wait at the end of this pass covers them)
        }
        __builtin_amdgcn_sched_barrier(0);
        const float* Xs = smem + buf * p.stage_floats;
        const float* Gs = Xs + p.g_floats_off;
        const int gr0 = fast_div(t0, p.inv_TW);
        const int pos0 = MO * (t0 - gr0 * p.TW);
        // this lane's tiles of the stage: t0 + kh, + 2, ..: (tile row, column) advance with carries instead of a division per pair.  (Tiles past
        // the split's end keep the pattern: their slots exist, their gradient pixels were staged as zeros.)
        int grow = fast_div(t0 + kh, p.inv_TW), tx = t0 + kh - grow * p.TW;
#pragma unroll
        for (int s = 0; s < TK / 2; ++s) {
            const int tk = 2 * s + kh;                     // this lane's tile of the pair
            const int sb = ((grow - gr0) * p.RW + MO * tx - pos0) * 32 + ch;
            tx += 2;
            if (tx >= p.TW) { tx -= p.TW; ++grow; }
            if (tx >= p.TW) { tx -= p.TW; ++grow; }
            float tc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) tc[j] = Xs[roa + sb + j * 32] + vs * Xs[rob + sb + j * 32];
            const float v0 = tc[0] - tc[2], v1 = tc[1] + tc[2], v2 = tc[2] - tc[1], v3 = tc[1] - tc[3];
#pragma unroll
            for (int hh = 0; hh < NBN; ++hh) {
                float zr[MO];
#pragma unroll
                for (int b = 0; b < MO; ++b) {
                    float z = za[0] * Gs[(0 * GP + tk * MO + b) * GC + 32 * hh + ch] + za[1] * Gs[(1 * GP + tk * MO + b) * GC + 32 * hh + ch];
                    if (MO == 3) z += za[2] * Gs[(2 * GP + tk * MO + b) * GC + 32 * hh + ch];
                    zr[b] = z;
                }
                float z0, z1, z2, z3;
                if (MO == 2) { z0 = zr[0]; z1 = zr[0] + zr[1]; z2 = zr[0] - zr[1]; z3 = -zr[1]; }
                else { z0 = zr[0]; z1 = zr[0] + zr[1] + zr[MO - 1]; z2 = zr[0] - zr[1] + zr[MO - 1]; z3 = zr[MO - 1]; }
                acc[hh][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(z0, v0, acc[hh][0], 0, 0, 0);
                acc[hh][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(z1, v1, acc[hh][1], 0, 0, 0);
                acc[hh][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(z2, v2, acc[hh][2], 0, 0, 0);
                acc[hh][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(z3, v3, acc[hh][3], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // nu side of G^T . G in registers:  MO = 2 (G 4x3): S0 = D0 + (D1 + D2)/2, S1 = (D1 - D2)/2, S2 = (D1 + D2)/2 + D3
    //                                   MO = 3 (G 4x2): S0 = D0 + (D1 + D2)/2, S1 = (D1 - D2)/2 - D3
    // D[row n = (e & 3) + 8 (e >> 2) + 4 (l >> 5)][col c = l & 31]
    float* out = p.part + ((((long long)split * phases + phase) * NS) * 4 + xi) * ((long long)p.CnPad * p.CxPad);
    const long long set_stride = 4LL * p.CnPad * p.CxPad;
#pragma unroll
    for (int hh = 0; hh < NBN; ++hh)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = n0 + 32 * hh + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const float d0 = acc[hh][0][e], d1 = acc[hh][1][e], d2 = acc[hh][2][e], d3 = acc[hh][3][e];
        float* o = out + (long long)row * p.CxPad + c0 + ch;
        o[0] = d0 + 0.5f * (d1 + d2);
        if (MO == 2) { o[set_stride] = 0.5f * (d1 - d2); o[2 * set_stride] = 0.5f * (d1 + d2) + d3; }
        else { o[set_stride] = 0.5f * (d1 - d2) - d3; }
    }
}

template <int MO, int TK, int NBN = 1>
__global__ __launch_bounds__(256, NBN == 2 ? 2 : 3) void winograd_wgrad_kernel(WinoWgradParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    winograd_wgrad_body<MO, TK, NBN>(p, smem);
}

// Sum of the split partials in split order + the xi side of G^T . G, written in the filter's own layout:
//   MO = 2: dw (Cout, Cin, 3, 3):  dw[n][c][r][s] = sum_xi G[xi][r] S_s[xi]      G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//   MO = 3: dw (Cin, Cout, 4, 4):  dw[c][n][3 - py - 2a][3 - px - 2b] = sum_xi G'[xi][a] S_b[xi]   G' = [1 0; .5 .5; .5 -.5; 0 -1]
// Sum of the split partials + the xi side of G^T . G, written in the filter's own layout:
//   MO = 2: dw (Cout, Cin, 3, 3):  dw[n][c][r][s] = sum_xi G[xi][r] S_s[xi]      G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//   MO = 3: dw (Cin, Cout, 4, 4):  dw[c][n][3 - py - 2a][3 - px - 2b] = sum_xi G'[xi][a] S_b[xi]   G' = [1 0; .5 .5; .5 -.5; 0 -1]
// One block per (phase, n, 64 consecutive c): thread = (c, split group g of 4); group g adds the splits sp = g, g + 4, .. in order, the four
// group sums are combined in a fixed order through LDS (deterministic), thread group 0 applies the xi side and stores.
template <int MO>
__global__ __launch_bounds__(256) void winograd_wgrad_finish_kernel(const float* __restrict__ part, float* __restrict__ dw, int Cn, int Cx, int CnPad,
                                                                     int CxPad, int splits) {
    constexpr int NS = MO == 2 ? 3 : 2, PH = MO == 2 ? 1 : 4;
    __shared__ float red[3][NS * 4][64];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int cblocks = (Cx + 63) / 64;
    int bid = blockIdx.x;
    const int cb = bid % cblocks; bid /= cblocks;
    const int n = bid % Cn;
    const int phase = bid / Cn;
    const int c = cb * 64 + cl;
    const long long plane = (long long)CnPad * CxPad;
    float s[NS][4];
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
        for (int x = 0; x < 4; ++x) s[k][x] = 0.f;
    if (c < Cx) {
        for (int sp = grp; sp < splits; sp += 4) {
            const float* q = part + (((long long)sp * PH + phase) * NS * 4) * plane + (long long)n * CxPad + c;
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int x = 0; x < 4; ++x) s[k][x] += q[(k * 4 + x) * plane];
        }
    }
    if (grp > 0) {
#pragma unroll
        for (int k = 0; k < NS; ++k)
#pragma unroll
            for (int x = 0; x < 4; ++x) red[grp - 1][k * 4 + x][cl] = s[k][x];
    }
    __syncthreads();
    if (grp == 0 && c < Cx) {
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int k = 0; k < NS; ++k)
#pragma unroll
                for (int x = 0; x < 4; ++x) s[k][x] += red[g][k * 4 + x][cl];
        if (MO == 2) {
            float* o = dw + ((long long)n * Cx + c) * 9;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                o[0 * 3 + k] = s[k][0] + 0.5f * (s[k][1] + s[k][2]);
                o[1 * 3 + k] = 0.5f * (s[k][1] - s[k][2]);
                o[2 * 3 + k] = 0.5f * (s[k][1] + s[k][2]) + s[k][3];
            }
        } else {
            const int py = phase >> 1, px = phase & 1;
            float* o = dw + ((long long)c * Cn + n) * 16;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int kx = 3 - px - 2 * k;
                o[(3 - py) * 4 + kx] = s[k][0] + 0.5f * (s[k][1] + s[k][2]);
                o[(1 - py) * 4 + kx] = 0.5f * (s[k][1] - s[k][2]) - s[k][3];
            }
        }
    }
}

static std::atomic<unsigned> g_ww_lds_done[4];
static std::atomic<int> g_ww_halves{2};            // vatl_tune_set(23, v): 32-channel gradient halves per block where the layer allows (1 or 2)
int wino_wgrad_set_halves(int v) { g_ww_halves.store(v, std::memory_order_relaxed); return 0; }

static std::atomic<int> g_ww_table{1};             // staging-address tables (vatl_tune_set(25, v): 0 = addresses computed in the kernel; identical results)
int wino_wgrad_set_table(int v) { g_ww_table.store(v, std::memory_order_relaxed); return 0; }

static std::atomic<int> g_ww_blocks{1024};         // target block count of a launch (vatl_tune_set(19, v))
int wino_wgrad_set_blocks(int v) { g_ww_blocks.store(v, std::memory_order_relaxed); return 0; }

constexpr int kWWTK = 8;
constexpr int kWWMaxLds = 64 * 1024;

struct WWPlan { int n_tiles, c_tiles, splits, tps, phases, ns, nbn; long long floats, part_floats, tab_x_words, tab_g_words; };

static WWPlan ww_plan(int MO, int Cn, int Cx, long long Mtiles, int TW) {
    WWPlan q{};
    q.ns = (kWWTK - 1) * MO + 4 + (4 - MO) * ((TW + kWWTK - 2) / TW);
    // two gradient halves per block where the channel count allows and both stages still fit the block's LDS
    // (measured at B = 120, tools/wino_wgrad_bench.py --halves: 3x3 layers with 256 / 512 channels +3 .. +10 %, 128 channels -3 %, the transposed convs
    // +-2 %: the second half's registers cost the third and fourth resident block, which is most of what the shared transform saves)
    int nbn = (g_ww_halves.load(std::memory_order_relaxed) >= 2 && MO == 2 && Cn % 64 == 0 && Cn >= 256 && Cx >= 256) ? 2 : 1;
    if (nbn == 2 && 2 * ((4 * q.ns * 8 + 63) / 64 + (MO * kWWTK * MO * 16 + 63) / 64) * 1024 > kWWMaxLds) nbn = 1;
    q.nbn = nbn;
    q.n_tiles = cdiv(Cn, 32 * nbn); q.c_tiles = cdiv(Cx, 32); q.phases = MO == 3 ? 4 : 1;
    const long long per_split = (long long)q.n_tiles * q.c_tiles * q.phases;
    // target block count: the knob for the 3x3 layers, 4x that for the transposed convs (B = 120, tools/wino_wgrad_bench.py: deconv3 1.03x
    // over the implicit GEMM at 1024 blocks, 1.13x at 2048, 1.16x at 4096; the 3x3 layers peak at 1024)
    long long want = std::max<long long>(1, (long long)g_ww_blocks.load(std::memory_order_relaxed) * (MO == 3 ? 4 : 1) / per_split);
    const long long stages = (Mtiles + kWWTK - 1) / kWWTK;
    want = std::min(want, stages);
    const long long sps = (stages + want - 1) / want;      // stages per split
    q.tps = (int)(sps * kWWTK);
    q.splits = (int)((Mtiles + q.tps - 1) / q.tps);
    const int NS = MO == 2 ? 3 : 2;
    q.floats = (long long)q.splits * q.phases * NS * 4 * (q.n_tiles * 32LL * nbn) * (q.c_tiles * 32LL);
    // + the staging-address tables (channel counts that are whole tiles only): x [phases][stages][ndma_x * 64], gradient [stages][ndma_g * 64]
    q.part_floats = q.floats;
    q.tab_x_words = q.tab_g_words = 0;
    if (Cx % 32 == 0 && Cn % (32 * nbn) == 0) {
        const long long ndx = (4 * q.ns * 8 + 63) / 64, ndg = (MO * kWWTK * MO * 8 * nbn + 63) / 64;
        q.tab_x_words = (long long)q.phases * stages * ndx * 64;
        q.tab_g_words = stages * ndg * 64;
        q.floats += q.tab_x_words + q.tab_g_words;
    }
    return q;
}

template <int MO>
static int winograd_wgrad_impl(const float* x, const float* g, float* dw, float* workspace, int N, int H, int W, int Cx, int Cn, hipStream_t st) {
    if (!x || !g || !dw || !workspace || N <= 0 || H <= 0 || W <= 0) return fail(VATL_EINVAL, "winograd_wgrad: null pointer or empty batch");
    if ((Cx & 3) || (Cn & 3)) return fail(VATL_EINVAL, "winograd_wgrad: channel counts %d / %d must be multiples of 4", Cx, Cn);
    WinoWgradParams p{};
    p.x = x; p.g = g; p.part = workspace;
    p.N = N; p.H = H; p.W = W; p.Cx = Cx; p.Cn = Cn;
    p.TH = (H + MO - 1) / MO; p.TW = (W + MO - 1) / MO; p.tpi = p.TH * p.TW;
    const long long mt = (long long)N * p.tpi;
    const int os = MO == 3 ? 2 : 1;
    const long long xe = (long long)N * H * W * Cx, ge = (long long)N * H * W * os * os * Cn;
    if (xe >= (1LL << 30) || ge >= (1LL << 30) || mt >= (1LL << 20) || (long long)N * p.TH >= (1LL << 20))
        return fail(VATL_EINVAL, "winograd_wgrad: tensor too large for this route (2^30 elements / 2^20 tiles); use the implicit GEMM");
    p.Mtiles = (int)mt;
    const WWPlan q = ww_plan(MO, Cn, Cx, mt, p.TW);
    p.n_tiles = q.n_tiles; p.c_tiles = q.c_tiles; p.splits = q.splits; p.tps = q.tps;
    const int nbn = q.nbn;
    p.CnPad = q.n_tiles * 32 * nbn; p.CxPad = q.c_tiles * 32;
    p.RW = MO * p.TW + 4 - MO; p.ns = q.ns;
    p.ndma_x = (4 * p.ns * 8 + 63) / 64;
    p.ndma_g = (MO * kWWTK * MO * 8 * nbn + 63) / 64;
    p.g_floats_off = p.ndma_x * 256;
    p.stage_floats = (p.ndma_x + p.ndma_g) * 256;
    if (p.ndma_x > 24 || p.RW > 4096 || p.TH > 4096) return fail(VATL_EINVAL, "winograd_wgrad: image %dx%d outside the range of this route", H, W);
    p.OH = H * os; p.OW = W * os; p.os = os; p.deconv = MO == 3;
    p.inv_TW = 1.0f / (float)p.TW; p.inv_TH = 1.0f / (float)p.TH; p.inv_RW = 1.0f / (float)p.RW; p.inv_ns = 1.0f / (float)p.ns;
    p.x_bytes = (unsigned)(xe * 4); p.g_bytes = (unsigned)(ge * 4);
    const int smem = 2 * p.stage_floats * (int)sizeof(float);
    if (smem > kWWMaxLds) return fail(VATL_EINVAL, "winograd_wgrad: %d bytes of LDS per block", smem);
    const dim3 grid((unsigned)((long long)q.n_tiles * q.c_tiles * q.phases * q.splits));
    p.stages_total = (int)((mt + kWWTK - 1) / kWWTK);
    if (q.tab_x_words > 0 && g_ww_table.load(std::memory_order_relaxed)) {
        unsigned* tx = reinterpret_cast<unsigned*>(workspace + q.part_floats);
        unsigned* tg = tx + q.tab_x_words;
        const dim3 tgrid((unsigned)p.stages_total, (unsigned)q.phases);
        if (nbn == 2) hipLaunchKernelGGL((winograd_wgrad_table_kernel<MO, kWWTK, 2>), tgrid, dim3(256), 0, st, p, tx, tg);
        else          hipLaunchKernelGGL((winograd_wgrad_table_kernel<MO, kWWTK, 1>), tgrid, dim3(256), 0, st, p, tx, tg);
        p.tab_x = tx; p.tab_g = tg;
        meter_route(kRouteWinoWgradTable);
    }
    if (nbn == 2) {
        auto kern = winograd_wgrad_kernel<MO, kWWTK, 2>;
        if (int rc = ensure_dynamic_lds((const void*)kern, kWWMaxLds, g_ww_lds_done[2 + MO - 2], "winograd_wgrad")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
    } else {
        auto kern = winograd_wgrad_kernel<MO, kWWTK, 1>;
        if (int rc = ensure_dynamic_lds((const void*)kern, kWWMaxLds, g_ww_lds_done[MO - 2], "winograd_wgrad")) return rc;
        hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, p);
    }
    meter_add(1, 2.0 * (double)p.CnPad * p.CxPad * 16.0 * q.phases * (double)((mt + kWWTK - 1) / kWWTK * kWWTK));
    meter_route(nbn == 2 ? kRouteWinoWgrad2H : kRouteWinoWgrad);
    if (int rc = check_launch("winograd_wgrad")) return rc;
    hipLaunchKernelGGL(winograd_wgrad_finish_kernel<MO>, dim3((unsigned)((long long)q.phases * Cn * ((Cx + 63) / 64))), dim3(256), 0, st, workspace, dw, Cn, Cx,
                       p.CnPad, p.CxPad, q.splits);
    return check_launch("winograd_wgrad_finish");
}

}  // namespace vatl

using namespace vatl;

extern "C" int64_t vatl_conv3x3_winograd_wgrad_workspace_floats(int Cout, int Cin, int64_t N, int H, int W) {
    return ww_plan(2, Cout, Cin, N * ((H + 1) / 2) * ((W + 1) / 2), (W + 1) / 2).floats;
}

extern "C" int vatl_conv3x3_winograd_wgrad(const float* x, const float* dz, float* dw, float* workspace, int N, int H, int W, int Cin, int Cout,
                                           void* stream) {
    return winograd_wgrad_impl<2>(x, dz, dw, workspace, N, H, W, Cin, Cout, (hipStream_t)stream);
}

extern "C" int64_t vatl_deconv4x4s2_winograd_wgrad_workspace_floats(int Cin, int Cout, int64_t N, int H, int W) {
    return ww_plan(3, Cout, Cin, N * ((H + 2) / 3) * ((W + 2) / 3), (W + 2) / 3).floats;
}

extern "C" int vatl_deconv4x4s2_winograd_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int H, int W, int Cin, int Cout,
                                               void* stream) {
    return winograd_wgrad_impl<3>(x, dy, dw, workspace, N, H, W, Cin, Cout, (hipStream_t)stream);
}
