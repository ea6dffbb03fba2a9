// Winograd convolutions on the gfx950 fp32 matrix cores:
//   * F(2x2, 3x3) for the 3x3 / stride 1 / pad 1 layers (MO = 2): a 2x2 output tile from its 4x4 input tile,
//   * F(3x3, 2x2) for ConvTranspose2d(4, 2, 1) (MO = 3): each of its four sub-pixel phases is a 2x2 convolution of the input; a 3x3
//     tile of phase outputs comes from a 4x4 input tile.  Its data gradient (a 4x4 / stride 2 conv over dz) is the sum over the four
//     PIXEL phases of dz of 2x2 convolutions: the same kernel with the reduction running over (phase, channel) (GATHER instantiation).
// Both need 16 multiplies per (tile, input channel, output channel) where the direct sums need 36: 2.25x fewer MFMAs.
//
// Why.  These layers are half of the R50 trunk's time and nearly all of HRNet's (profiles/r03_layer_report.txt), and the implicit GEMM
// already runs them at 76 - 89 % of the fp32 MFMA peak: the only way to make them substantially faster is fewer multiplies.
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A          U = G g G^T (16 values per filter, packed once: vatl_pack_winograd_*_weight)
//                                                   V = B^T d B (16 values per tile and input channel, computed on the fly)
//
// i.e. 16 independent GEMMs  M_p[tile][n] = sum_c V_p[tile][c] U_p[c][n]  (p = (xi, nu), the position in the 4x4 transform domain), still
// exact fp32 products with fp32 accumulation (cuDNN picks the same algorithm for these layers of the reference: the rounding differs
// from the direct sum in the last bits, not in class; tests hold it to the same tolerance).  With the interpolation points (0, 1, -1,
// inf) the two variants share B^T (rows d0 - d2, d1 + d2, d2 - d1, d1 - d3) — the input path of the kernel is the same — and differ in
// the tile step (MO), in G (4x3 / 4x2) and in A^T (2x4: [1 1 1 0; 0 1 -1 -1] / 3x4: [1 1 1 0; 0 1 -1 0; 0 1 1 1]).
//
// Block = 32 tiles (flat index over image, tile row, tile column) x 32 output channels x all 16 positions; 4 waves, 64 accumulator
// registers per lane, three blocks per CU (12 waves): the blocks of a CU overlap each other's prologue, barriers and output transform.
// Wave xi owns the four positions (xi, nu = 0..3).  (64-tile blocks with 128 accumulator registers — 8 waves x 64 channels, one block
// per CU, or 4 waves x 32 channels, two per CU — measured 4 - 11 % slower at 1024 crops and 25 - 35 % slower at 120: removed.  FOUR blocks
// per CU — 128 VGPRs with a single filter-fragment set loaded just in time — 5 - 12 % slower (3x3) / 25 - 60 % slower (transposed convs): the
// one-step look-ahead of the filter loads is worth more than the fourth block.  A TWO-step look-ahead (three fragment sets) needs 171
// VGPRs, i.e. 13 spills under the 168 of three waves per SIMD: 5 - 15 % slower.  s_setprio around the MFMA bursts: no effect.)
//   * B operand (U): packed in MFMA fragment order, a wave's 16-byte-per-lane load is 1 KB contiguous; straight from L2 into registers
//     one 8-channel step ahead (no LDS: every wave needs a different slice).
//   * A operand (V): the raw input pixels of the block's tiles are staged in LDS 16 channels at a time by LDS-DMA (double buffer, one
//     stage ahead), de-duplicated along the tile row: per input row i of the tiles, MO column arrays — column j of tile t is entry
//     t + j / MO of array j % MO, with a FLAT tile index, so neighbouring tiles share their 4 - MO common columns and consecutive tiles
//     are consecutive 64-byte entries (bank-conflict-free ds_read_b128 for the hardware's lane groups; the details are at the loader).
//     A wave needs two of the four tile rows (row transform) and forms its four V_p = column transform in registers: 8 ds_read_b128
//     and 8 vector adds per 16 MFMAs.  The 16-byte read of a lane is four consecutive channels = four MFMA k-steps (the same k
//     permutation on both operands).
//   * Output transform: the nu sum happens in registers (4 accumulator tiles -> MO), the xi sum through LDS (the staging buffers are
//     free by then): every wave writes its partial tiles, every thread then combines the four xi for whole 16-byte channel groups and
//     stores NHWC channel runs, with scale / bias / residual / ReLU, the BatchNorm statistics of the training forward, or the
//     BatchNorm-backward reduction of the fine-tune step's data gradients fused.
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
    int N, H, W, Cin, Cout;           // H x W: the input image = the grid the tiles cover (conv output / one deconv phase)
    int TH, TW, tpi, Mtiles;          // tiles per image column / row / image, tiles in the launch
    int m_tiles, n_tiles;
    int relu;
    int stages;                       // Cin / 16
    int nhp;                          // 32-channel groups per filter tile of the packing (1 if Cout <= 32, else 2)
    int pad_y, pad_x;                 // tile (ty, tx) reads input rows MO ty - pad_y + i (deconv: 1 - phase bit, set in the kernel)
    int OH, OW, os, ooy, oox;         // output pixel of grid point (y, x) = (y os + ooy, x os + oox) in an OH x OW image
    int deconv;                       // four sub-pixel phases py * 2 + px, each with its own filter
    int gather;                       // data gradient of the transposed conv: the reduction runs over (input phase, channel); the pixels of
                                      // phase (p, q) are x[2 y + p][2 x + q] of an (N, 2H, 2W, Cin) tensor, read with pad (p, q)
    int spp;                          // gather: 16-channel stages per phase (Cin / 16); stages = 4 spp
    int rn;                           // filter slices per group of the tile order (see the kernel)
    // persistent route (winograd_persist_kernel): the launch covers `periods` runs of `pg` 32-tile groups = lcm(tiles per image, 32) tiles =
    // a whole number of images, so that the geometry of a group repeats from period to period; a block keeps one (group, filter tile) of the
    // period and walks `chunk` consecutive periods, advancing its buffer bases by the period's bytes
    int pg, periods, chunk;
    long long x_period_floats, y_period_floats;
    long long u_phase_floats;         // deconv: floats of one phase's packed filter
    float neg_one;                    // -1.0f, as a run-time value: a - b is written fma(b, neg_one, a), which hipcc packs two lanes per instruction (a literal -1 is folded back
                                      // into four scalar v_sub_f32; vector instructions are paid in full next to the MFMAs: profiles/r04_notes.md).  Exact: same bits.
    int ablate;                       // profiling library only (vatl_tune_set(17, bits), wrong results): 1 no output transform, 2 no LDS
                                      // reads / input transform, 4 no filter loads, 8 no staging DMA, 16 no barriers
    unsigned x_bytes, u_bytes, y_bytes;
    // divisions by launch constants (common.h: fdiv): a block's setup has ~30 of them per lane, which for the short layers (Cin = 32 / 64:
    // 64 / 128 MFMAs per wave and block) cost as much as the MFMAs when done as integer divisions
    FastDivU d_TH, d_TW, d_tpi, d_grp, d_rn, d_rn_last, d_ntiles;
};

constexpr unsigned WOOB_BASE = 0xF0000000u;               // out of range for every descriptor, and still so with a fragment offset added
constexpr unsigned WOOB = 0xFFFFFFFFu;
constexpr unsigned WOOB_G = 0xFFFF0000u;                  // staging offset of a zero piece: still out of range with stage * 64 bytes added (stages < 1024, tensors <= 0xFFFF0000 bytes: checked on the host), so the per-stage offset needs no select
typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 wbuf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ void wbuf_store4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(wu32x4, v), r, byte_off, 0, 0);
}

constexpr int W_TB = 32;                        // tiles per block
constexpr int W_CK = 16;                        // channels per LDS stage
constexpr int W_NQ = W_TB + 1;                  // entries of a column array: one per tile + the halo of the last tile
constexpr int W_ZERO = 16;                      // floats of the zero pixel in front of the stages
constexpr int W_LDP = 32;                       // row stride of the output-transform tiles in LDS
// LDS stage: [4 input rows i][MO column arrays r][33 entries q][16 channels]; entry (r, q) = column r of tile q (= column MO + r of tile q - 1)
template <int MO> struct WinoStage {
    static constexpr int ROWE = MO * W_NQ + 1;             // entries per input row: MO column arrays of W_NQ + ONE ZERO ENTRY (its four pieces are requested out of range by
                                                           // every stage): the column of a tile that lies outside the image reads it — at the same index in every row, so the
                                                           // fragment reads need no per-lane select (2 per column and step before; vector instructions are not hidden behind the MFMAs)
    static constexpr int ITEMS = 4 * ROWE * 4;             // 16-byte pieces
    static constexpr int NDMA = (ITEMS + 63) / 64;         // wave-wide LDS-DMA instructions (1 KB each; the last one is partly used)
    static constexpr int FLOATS = NDMA * 256;
    static constexpr int NLD = (NDMA + 3) / 4;             // per wave
};

typedef __attribute__((address_space(3))) void wlds_void;

// (body in a __device__ function: with the DMA builtin inside the __global__ template hipcc 7.2 drops the kernel's host stub)
template <int MO, bool BNB, bool GATHER, int NB>
__device__ __forceinline__ void winograd_body(const WinoParams& p, float* smem, const int bid, const int nblk) {
    constexpr int NT = 256, BN = 32 * NB, NW = 4;       // NB: 32-channel halves of the filter tile per block (2: the staged pixels and their transform serve 64 output channels)
    using ST = WinoStage<MO>;
    constexpr int W_NLD = ST::NLD, STAGE = ST::FLOATS, ROWF = ST::ROWE * W_CK;   // DMA instructions per wave, floats per stage / per input row
    float* Rs = smem + W_ZERO;                             // [2][4 rows][MO arrays][33 entries][16 channels], chunk-swizzled

    int tid_ = threadIdx.x;
    if constexpr (GATHER) asm volatile("" : "+v"(tid_));   // (part of the gather instantiations' loop form, see winograd_kernel)
    const int tid = tid_, lane = tid & 63, xi = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware tile order: block b runs on XCD b % 8; each XCD gets a contiguous run of the sequence
    //     for (group of rn filter slices) for (m-tile) for (slice in the group)        slice = (phase, 32 output channels)
    // so the blocks that are resident together read the same input tiles (fetched from HBM once per group instead of once per slice)
    // while the group's filter slices (rn * Cin * 2 KB <= 2 MB) stay in that XCD's L2 for the whole sweep over the m-tiles.
    const int xcd = bid & 7, loc = bid >> 3, q8 = nblk >> 3, r8 = nblk & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int units = p.n_tiles * (p.deconv ? 4 : 1);      // (gather mode: deconv = 0, the four input phases are part of the reduction)
    const int grp = fdiv(t, p.d_grp), rem = t - grp * (p.m_tiles * p.rn);
    const bool last_grp = units - grp * p.rn < p.rn;       // the last group may be smaller
    const int rn_g = last_grp ? units - grp * p.rn : p.rn;
    const int m_tile = fdiv(rem, last_grp ? p.d_rn_last : p.d_rn), unit = grp * p.rn + (rem - m_tile * rn_g);
    const int phase = fdiv(unit, p.d_ntiles), n_tile = unit - phase * p.n_tiles;
    const int m0 = m_tile * W_TB, n0 = n_tile * BN;

    int pad_y = p.pad_y, pad_x = p.pad_x, ooy = p.ooy, oox = p.oox;
    const float* ubase_ptr = p.u;
    if (p.deconv) {
        const int py = phase >> 1, px = phase & 1;
        pad_y = 1 - py; pad_x = 1 - px; ooy = py; oox = px;
        ubase_ptr += (long long)phase * p.u_phase_floats;
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ubase_ptr), 0, p.u_bytes, 0x00020000);

#ifdef VATL_ABLATION
    const int abl = p.ablate;
#else
    constexpr int abl = 0;
#endif
    // The first requests of the block go out as early as their addresses exist (the first filter fragments here, the first stage right
    // after its offsets, the write-out's operands before the accumulators go through LDS): a block with two to four stages (32 / 64
    // input channels) is not much longer than these latencies.  Not in the gather instantiations: their blocks run 64 - 128 stages and
    // they are short of registers as it is.
    constexpr bool EARLY = !GATHER;
    f32x4 ua[4], ub[4];
    if constexpr (EARLY) {
        const int hn0 = n_tile * NB, ut0 = hn0 / p.nhp;
        const unsigned u0 = (unsigned)((((ut0 * p.stages * 2) * 16 + 4 * xi) * p.nhp + (hn0 - ut0 * p.nhp)) * 64 + lane) << 4;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) ua[nu] = wbuf_load4(ur, (abl & 4) ? WOOB : u0 + nu * (p.nhp * 1024u));
    }

    // ---- staging by LDS-DMA (buffer_load ... lds: no staging registers, no ds_write pass).  The destination of a wave instruction is
    // lane-linear (base + lane * 16 bytes), so the chunk swizzle is applied on the SOURCE side: LDS position p = (entry p >> 2, chunk
    // position p & 3) receives the entry's global chunk (p & 3) ^ ((q >> 2) & 3).
    // Column arrays: pixel column j of tile t (x = MO tx + j - pad) lives in array r = j % MO at entry q = t + j / MO — the tile index
    // is FLAT, so consecutive tiles are consecutive 64-byte entries even across tile-row ends and the 16 lanes of a ds_read_b128 group
    // ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}) hit 16 different 16-byte bank slots (with one slot per distinct pixel column —
    // stride MO between tiles and a gap at row ends — every fragment read was a 3-way bank conflict: SQ_LDS_BANK_CONFLICT 65 % of the
    // LDS cycles).  At a row end entry (r, q) is asked for by two tiles: column r of tile q (first of its row: x = r - pad) and
    // column MO + r of tile q - 1 (last of its row: x = MO TW + r - pad) — never both inside the image; the entry holds whichever is,
    // and a tile whose column is outside the image reads the zero pixel instead.
    unsigned goff[W_NLD];                                  // byte offset of the lane's piece in the first stage (WOOB_G: zeros)
    auto set_goff = [&](int ph) {                          // ph: input phase of the gather mode (0 otherwise)
        const int gy = ph >> 1, gx = ph & 1;
        const int py_ = GATHER ? gy : pad_y, px_ = GATHER ? gx : pad_x;
#pragma unroll
        for (int u = 0; u < W_NLD; ++u) {
            const int pz = (xi + NW * u) * 64 + lane;
            goff[u] = WOOB_G;
            if (pz >= ST::ITEMS) continue;
            const int cpos = pz & 3, e = pz >> 2;
            const int i = e / ST::ROWE, re = e - i * ST::ROWE;
            const int r = re / W_NQ, q = re - r * W_NQ;
            if (r >= MO) continue;                         // the row's zero entry
            const int chunk = cpos ^ ((q >> 2) & 3);
            int m = m0 + q;                                // column r of tile m ...
            int gr = fdiv(m, p.d_TW), tx = m - gr * p.TW;
            int xx = MO * tx + r - px_;
            if (m >= p.Mtiles || (tx == 0 && xx < 0)) {    // ... or, when that is outside, column MO + r of the tile before (a row end / the halo)
                m -= 1;
                if (m < 0) continue;
                gr = fdiv(m, p.d_TW); tx = m - gr * p.TW;
                xx = MO * tx + MO + r - px_;
            }
            const int b = fdiv(gr, p.d_TH), ty = gr - b * p.TH;
            const int yy = MO * ty - py_ + i;
            if (m < p.Mtiles && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W) {
                if (GATHER) goff[u] = (unsigned)(((b * 2 * p.H + 2 * yy + gy) * 2 * p.W + 2 * xx + gx) * p.Cin + chunk * 4) << 2;
                else goff[u] = (unsigned)(((b * p.H + yy) * p.W + xx) * p.Cin + chunk * 4) << 2;
            }
        }
    };
    set_goff(0);
    auto stage_dma = [&](int buf, int st) {
        int cst = st;                                      // 16-channel slice inside the (phase's) tensor
        if constexpr (GATHER) {
            const int ph = st / p.spp;
            cst = st - ph * p.spp;
            if (cst == 0 && ph > 0) set_goff(ph);          // a new input phase: other pixels, other padding
        }
#pragma unroll
        for (int u = 0; u < W_NLD; ++u)
            if (xi + NW * u < ST::NDMA)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (wlds_void*)(Rs + buf * STAGE + (xi + NW * u) * 256), 16,
                                                         goff[u] + (unsigned)cst * (W_CK * 4), 0, 0, 0);
    };

    if constexpr (EARLY) {
        if (tid < W_ZERO / 4) *reinterpret_cast<f32x4*>(&smem[tid * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};
        stage_dma(0, 0);                                   // in flight while the rest of the prologue runs
        // ... and the second stage with it (its buffer is free: nothing has been staged yet).  Requested inside stage 0 it had one
        // group of MFMAs to land, and the blocks of the short layers (32 / 64 input channels: two to four stages) waited for it a second
        // time at the end of that stage: one memory latency less on the serial chain of every block.
        if (p.stages > 1 && !(abl & 8)) stage_dma(1, 1);
    }

    // ---- fragment addressing: lane = (tile l & 31, channel quad l >> 5).  ra[px][j]: float index (relative to Rs, input row 0, stage 0,
    // first 8-channel step) of column j of the lane's tile; the second step of a stage is the same index ^ 8; a column outside the image = the row's zero entry
    // (column outside the image).  The gather mode has two horizontal paddings (input phase & 1).
    const int h = lane >> 5;
    int ra[GATHER ? 2 : 1][4];
    {
        const int tl = lane & 31;
        const int m = min(m0 + tl, p.Mtiles - 1);
        const int gr = fdiv(m, p.d_TW), tx = m - gr * p.TW;
#pragma unroll
        for (int v = 0; v < (GATHER ? 2 : 1); ++v) {
            const int px_ = GATHER ? v : pad_x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = MO * tx + j - px_;
                const int q = tl + j / MO;
                ra[v][j] = (unsigned)xx < (unsigned)p.W ? ((j % MO) * W_NQ + q) * W_CK + ((h ^ ((q >> 2) & 3)) << 2) : MO * W_NQ * W_CK + (h << 2);
            }
        }
    }
    // row transform of this wave: t = d[ia] + sgn * d[ib]   (B^T rows: d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    const int ia = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int ib = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float sgn = xi == 1 ? 1.f : -1.f;
    const f32x4 m1 = {p.neg_one, p.neg_one, p.neg_one, p.neg_one};
    const int roa = ia * ROWF, rob = ib * ROWF;

    // ---- U fragments: [n_tile][step][position][nh][lane][4]; the packing groups 32 p.nhp channels per filter tile ----------------------
    const int steps = p.stages * 2;
    const int hn = n_tile * NB;                            // first 32-channel half of this block's filter tile (NB = 2: the second one follows 1 KB later)
    const int ut = hn / p.nhp, nh_g = hn - ut * p.nhp;
    const int usteps = GATHER ? 2 * p.spp : steps;        // steps of one packed filter
    const unsigned ubase = (unsigned)((((ut * usteps) * 16 + 4 * xi) * p.nhp + nh_g) * 64 + lane) << 4;   // bytes; + step * ustep + nu * unu
    const unsigned ustep = 16u * p.nhp * 1024u, unu = p.nhp * 1024u;
    const unsigned uphase = (unsigned)(p.u_phase_floats * 4);   // gather: bytes between the filters of consecutive input phases
    // (a request past the last step re-reads the last step's fragments instead of being switched off: the look-ahead of the final stage costs no
    // per-lane select — vector instructions are paid in full next to the MFMAs, profiles/r04_notes.md)
    auto u_load = [&](f32x4 (&dst)[4], int step_) {
        const int step = min(step_, steps - 1);
        unsigned off = ubase + (unsigned)step * ustep;
        if constexpr (GATHER) {
            const int ph = step / (2 * p.spp);
            off = ubase + (unsigned)ph * uphase + (unsigned)(step - ph * 2 * p.spp) * ustep;
        }
        if (abl & 4) off = WOOB_BASE;                      // (profiling ablation; abl is 0 in the product build)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) dst[nu] = wbuf_load4(ur, off + nu * unu);
    };

    f32x16 accs[NB][4];
#pragma unroll
    for (int hh = 0; hh < NB; ++hh)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int e = 0; e < 16; ++e) accs[hh][nu][e] = 0.f;
    f32x16 (&acc)[4] = accs[0];

    if constexpr (!EARLY) {
        if (tid < W_ZERO / 4) *reinterpret_cast<f32x4*>(&smem[tid * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};
        stage_dma(0, 0);
        u_load(ua, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's DMA pieces have landed
    __syncthreads();

    auto make_v = [&](f32x4 (&v)[4], const float* Rb, int x8, int pv) {      // V = B^T d B of the lane's tile: this wave's four positions, four channels
        f32x4 tc[4];
        if (abl & 2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) tc[j] = f32x4{1.f + j, 2.f, 3.f, 4.f + x8};
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int a = ra[GATHER ? pv : 0][j];
#ifdef VATL_ABLATION
                if (abl & 32) {        // lane-linear fragment addresses (bank-conflict probe)
                    const f32x4 da = *reinterpret_cast<const f32x4*>(Rb + lane * 4 + j * 256);
                    const f32x4 db = *reinterpret_cast<const f32x4*>(Rb + lane * 4 + j * 256 + 1024);
                    tc[j] = da + sgn * db;
                    continue;
                }
#endif
                const f32x4 da = *reinterpret_cast<const f32x4*>(Rb + roa + (a ^ x8));
                const f32x4 db = *reinterpret_cast<const f32x4*>(Rb + rob + (a ^ x8));
                tc[j] = da + sgn * db;
            }
        }
        v[0] = tc[0] + m1 * tc[2]; v[1] = tc[1] + tc[2]; v[2] = tc[2] + m1 * tc[1]; v[3] = tc[1] + m1 * tc[3];
    };
    auto mfma_group = [&](f32x16 (&ac)[4], const f32x4 (&v)[4], const f32x4 (&uu)[4]) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
                ac[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu][tt], uu[nu][tt], ac[nu], 0, 0, 0);
    };
    auto step_mfma = [&](const float* Rb, int x8, const f32x4 (&uu)[4], int pv) {
        f32x4 v[4];
        make_v(v, Rb, x8, pv);
        mfma_group(acc, v, uu);
    };

    if constexpr (NB == 2) {
        // Two filter halves per block: one V per step serves both, the two fragment sets alternate between the halves with a look-ahead of
        // one group of 16 MFMAs: (step 0, half 0) = ua [requested in the previous stage], (step 0, half 1) = ub, (step 1, half 0) = ua, ..
        constexpr bool V_AHEAD = !GATHER && (MO == 2 || !BNB);   // 16 more registers: the gather instantiations and F(3x3,2x2) with the BatchNorm-backward epilogue have none to spare
        auto u_load_h = [&](f32x4 (&dst)[4], int step_, int half) {
            const int step = min(step_, steps - 1);
            unsigned off = ubase + (unsigned)half * 1024u + (unsigned)step * ustep;
            if constexpr (GATHER) {
                const int ph = step / (2 * p.spp);
                off = ubase + (unsigned)half * 1024u + (unsigned)ph * uphase + (unsigned)(step - ph * 2 * p.spp) * ustep;
            }
            if (abl & 4) off = WOOB_BASE;
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) dst[nu] = wbuf_load4(ur, off + nu * unu);
        };
        for (int st = 0; st < p.stages; ++st) {
            const int buf = st & 1;
            const float* Rb = Rs + buf * STAGE;
            const int pv = GATHER ? (st / p.spp) & 1 : 0;
            f32x4 v[4], v1[4];
            u_load_h(ub, 2 * st, 1);
            __builtin_amdgcn_sched_barrier(0);
            make_v(v, Rb, 0, pv);
            mfma_group(accs[0], v, ua);
            if constexpr (V_AHEAD) make_v(v1, Rb, 8, pv);  // the second step's V in the shadow of the first group's MFMAs (no scheduling barrier in between)
            __builtin_amdgcn_sched_barrier(0);
            u_load_h(ua, 2 * st + 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(accs[NB - 1], v, ub);
            __builtin_amdgcn_sched_barrier(0);
            u_load_h(ub, 2 * st + 1, 1);
            if (st + 1 < p.stages && (!EARLY || st > 0) && !(abl & 8)) stage_dma(buf ^ 1, st + 1);     // (behind the fragments of this stage's second step: loads retire in order)
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (V_AHEAD) {
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) v[nu] = v1[nu];
            } else make_v(v, Rb, 8, pv);
            mfma_group(accs[0], v, ua);
            __builtin_amdgcn_sched_barrier(0);
            u_load_h(ua, 2 * st + 2, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(accs[NB - 1], v, ub);
            __builtin_amdgcn_sched_barrier(0);
            if (!(abl & 16)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }
    } else
    for (int st = 0; st < p.stages; ++st) {
        const int buf = st & 1;
        const float* Rb = Rs + buf * STAGE;
        const int pv = GATHER ? (st / p.spp) & 1 : 0;     // horizontal padding of this stage's input phase
        // the scheduler barriers keep the requests where they are written: without them hipcc sinks the filter loads to just before
        // their first MFMA (latency fully exposed)
        // (the filter loads first: VMEM loads retire in order, so fragments requested AFTER the staging DMA would make the second step wait
        // for the whole DMA; this way the DMA has both steps to land)
        u_load(ub, 2 * st + 1);
        if (st + 1 < p.stages && (!EARLY || st > 0) && !(abl & 8)) stage_dma(buf ^ 1, st + 1);
        __builtin_amdgcn_sched_barrier(0);
        step_mfma(Rb, 0, ua, pv);
        __builtin_amdgcn_sched_barrier(0);
        u_load(ua, 2 * st + 2);
        __builtin_amdgcn_sched_barrier(0);
        step_mfma(Rb, 8, ub, pv);
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
            for (int e = 0; e < 16; ++e) sacc += acc[nu][e];
        if (sacc == 12345.678f) p.y[0] = sacc;
        return;
    }
#endif
    constexpr int C4 = 32 / 4;                             // channel quads per tile row of one 32-channel half
#pragma unroll
    for (int hh = 0; hh < NB; ++hh) {                      // one filter half at a time through the same LDS tiles
        const int n0h = n0 + 32 * hh;
        f32x16 (&acch)[4] = accs[hh];
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
        // ---- output transform -----------------------------------------------------------------------------------------------------------------
        // What the write-out needs from global memory (store offsets, skip-connection values, per-channel constants) is requested FIRST, so
        // that it is in flight while the accumulators go through LDS: at the end of the block these latencies are not hidden by anything else.
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.res ? p.y_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNB ? p.bz : p.x), 0, BNB ? p.y_bytes : 0u, 0x00020000);
        const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BNB ? p.bmy : p.x), 0, (BNB && p.bmy) ? p.y_bytes : 0u, 0x00020000);
        const int c4 = tid % C4;
        const int n = n0h + c4 * 4;
        const bool nv = n < p.Cout;
        const f32x4 one = {1.f, 1.f, 1.f, 1.f}, nul = {0.f, 0.f, 0.f, 0.f};
        const float lo = p.relu ? 0.f : -INFINITY;
        f32x4 sc = one, bi = nul, mu = nul, is = nul, msc = nul, mbi = one;       // no mask: 0 * z + 1 > 0
        auto request_consts = [&]() {
            if (nv && p.scale) sc = *reinterpret_cast<const f32x4*>(p.scale + n);
            if (nv && p.bias) bi = *reinterpret_cast<const f32x4*>(p.bias + n);
            if (BNB && nv) {
                mu = *reinterpret_cast<const f32x4*>(p.bmu + n); is = *reinterpret_cast<const f32x4*>(p.bis + n);
                if (p.bsc) { msc = *reinterpret_cast<const f32x4*>(p.bsc + n); mbi = *reinterpret_cast<const f32x4*>(p.bbi + n); }
            }
        };
        unsigned off[MO][MO];                                  // [pass u][output row a] of the thread's (tile, output column, channel quad)
        f32x4 rs[MO][MO];
        constexpr bool EARLY_Z = BNB && MO == 2;               // the BatchNorm-backward operands too, where the registers allow (F(3x3,2x2): 3 x 36 more)
        f32x4 zq[EARLY_Z ? MO : 1][EARLY_Z ? MO : 1], yq[EARLY_Z ? MO : 1][EARLY_Z ? MO : 1];
        auto request_out = [&](const int u) {                  // pass u: store offsets, skip-connection values (BatchNorm-backward operands)
        {
            const int rest = (tid + NT * u) / C4;              // 0 .. 32 MO - 1
            const int tl = rest / MO, bq = rest - tl * MO;
            const int m = m0 + tl;
    #pragma unroll
            for (int a = 0; a < MO; ++a) off[u][a] = WOOB;
            if (nv && m < p.Mtiles) {
                const int b = fdiv(m, p.d_tpi), r = m - b * p.tpi;
                const int ty = fdiv(r, p.d_TW), tx = r - ty * p.TW;
                const int xx = MO * tx + bq;
                if (xx < p.W) {
    #pragma unroll
                    for (int a = 0; a < MO; ++a) {
                        const int yy = MO * ty + a;
                        if (yy < p.H) off[u][a] = (unsigned)(((b * p.OH + yy * p.os + ooy) * p.OW + xx * p.os + oox) * p.Cout + n) << 2;
                    }
                }
            }
    #pragma unroll
            for (int a = 0; a < MO; ++a) rs[u][a] = p.res ? wbuf_load4(rr, off[u][a]) : nul;
            if constexpr (EARLY_Z) {
    #pragma unroll
                for (int a = 0; a < MO; ++a) { zq[u][a] = wbuf_load4(zr, off[u][a]); yq[u][a] = p.bmy ? wbuf_load4(mr, off[u][a]) : nul; }
            }
        }
        };
        if constexpr (EARLY) {
            request_consts();
    #pragma unroll
            for (int u = 0; u < MO; ++u) request_out(u);
            __builtin_amdgcn_sched_barrier(0);
        }

        // nu sum in registers (rows of A^T):  MO = 2: P0 = M0 + M1 + M2, P1 = M1 - M2 - M3;   MO = 3: P0 = M0 + M1 + M2, P1 = M1 - M2,
        // P2 = M1 + M2 + M3.   Ps[xi][b][tile][n] in LDS
        float* Ps = smem;
        {
            const int cl = lane & 31;
    #pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const float m0v = acch[0][e], m1v = acch[1][e], m2v = acch[2][e], m3v = acch[3][e];
                Ps[((xi * MO + 0) * W_TB + row) * W_LDP + cl] = m0v + m1v + m2v;
                if (MO == 2) {
                    Ps[((xi * MO + 1) * W_TB + row) * W_LDP + cl] = m1v - m2v - m3v;
                } else {
                    Ps[((xi * MO + 1) * W_TB + row) * W_LDP + cl] = m1v - m2v;
                    Ps[((xi * MO + MO - 1) * W_TB + row) * W_LDP + cl] = m1v + m2v + m3v;
                }
            }
        }
        lds_barrier();                                     // (LDS hand-off only: the stores of the half before and the skip-connection loads requested above stay in flight)
        if constexpr (!EARLY) request_consts();

        // xi sum (the same rows of A^T) per (tile, output column b, channel quad); 16-byte stores of NHWC channel runs
    #pragma unroll
        for (int u = 0; u < MO; ++u) {
            if constexpr (!EARLY) request_out(u);
            const int rest = (tid + NT * u) / C4;
            const int tl = rest / MO, bq = rest - tl * MO;
            f32x4 pq[4];
    #pragma unroll
            for (int k = 0; k < 4; ++k) pq[k] = *reinterpret_cast<const f32x4*>(&Ps[((k * MO + bq) * W_TB + tl) * W_LDP + c4 * 4]);
            f32x4 yv[MO];
            yv[0] = pq[0] + pq[1] + pq[2];
            if (MO == 2) {
                yv[1] = pq[1] - pq[2] - pq[3];
            } else {
                yv[1] = pq[1] - pq[2];
                yv[MO - 1] = pq[1] + pq[2] + pq[3];
            }
            if constexpr (BNB) {
                f32x4 zt[MO], yt[MO];
    #pragma unroll
                for (int a = 0; a < MO; ++a) {
                    if constexpr (EARLY_Z) { zt[a] = zq[u][a]; yt[a] = yq[u][a]; }
                    else { zt[a] = wbuf_load4(zr, off[u][a]); yt[a] = p.bmy ? wbuf_load4(mr, off[u][a]) : nul; }
                }
    #pragma unroll
                for (int a = 0; a < MO; ++a) {
                    f32x4 gq;
    #pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float d = yv[a][c] + rs[u][a][c];
                        const bool on = p.bmy ? yt[a][c] > 0.f : fmaf(zt[a][c], msc[c], mbi[c]) > 0.f;
                        gq[c] = (on && off[u][a] != WOOB) ? d : 0.f;
                        ssum[c] += gq[c];
                        ssq[c] += gq[c] * ((zt[a][c] - mu[c]) * is[c]);
                    }
                    wbuf_store4(yr, off[u][a], gq);
                }
            } else {
    #pragma unroll
                for (int a = 0; a < MO; ++a) {
                    f32x4 o;
    #pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = fmaxf(yv[a][c] * sc[c] + bi[c] + rs[u][a][c], lo);
                    wbuf_store4(yr, off[u][a], o);
                    if (p.stats && off[u][a] != WOOB) {
    #pragma unroll
                        for (int c = 0; c < 4; ++c) { ssum[c] += o[c]; ssq[c] += o[c] * o[c]; }
                    }
                }
            }
        }
        if (p.stats) {
            // BatchNorm batch statistics of the pixels just stored: one (sum, sum^2) double pair per (row block, channel), fixed order
            lds_barrier();
            f32x4* sh = reinterpret_cast<f32x4*>(smem);
            sh[tid] = ssum; sh[NT + tid] = ssq;
            lds_barrier();
            if (tid < C4) {
                double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
    #pragma unroll 2
                for (int k = 0; k < NT / C4; ++k) {
                    const f32x4 a = sh[k * C4 + tid], b = sh[NT + k * C4 + tid];
    #pragma unroll
                    for (int c = 0; c < 4; ++c) { ds[c] += a[c]; dq[c] += b[c]; }
                }
                const long long rb = (long long)phase * p.m_tiles + m_tile;
    #pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int nn = n0h + tid * 4 + c;
                    if (nn < p.Cout) {
                        p.stats[(rb * p.Cout + nn) * 2 + 0] = ds[c];
                        p.stats[(rb * p.Cout + nn) * 2 + 1] = dq[c];
                    }
                }
            }
        }
        if (NB > 1 && hh + 1 < NB) lds_barrier();          // the next half's tiles go where this one's were read (this half's stores stay in flight)
    }
}

// Persistent variant for the 3x3 layers (MO = 2) whose blocks are short — 32 .. 128 input channels: two to eight stages — i.e. every
// branch conv of HRNet and the first stages of the ResNets.  Measured on those layers (profiles/r04_notes.md): the tensors can sit in the
// Infinity Cache or stream from HBM, with or without the skip-connection read, and a launch takes the same time — what the blocks spend
// beside their MFMAs is INSTRUCTIONS: ~900 before the first MFMA (tile decode, ~30 magic-number divisions per lane for the staging and store
// offsets) and ~700 after the last one, for 64 .. 256 MFMAs per wave.  All of that geometry depends on the tile's position inside its image
// only.  With a launch cut into periods of lcm(tiles per image, 32) tiles (a whole number of images AND of 32-tile groups) group g of every
// period has the same geometry, so a block computes it ONCE — staging offsets, fragment addresses, store offsets, masks — keeps it in
// registers and walks the periods by advancing the base addresses of its buffer descriptors (scalar work).  Per-tile arithmetic, stage
// order and summation order are those of winograd_body: bit-identical results.
// PF (two blocks per CU): the output-transform tiles get their own LDS region instead of lying over the stage buffers, so the first two stages and
// the first filter fragments of the NEXT period are requested right after the last stage of this one and land while its write-out runs; without
// it every period starts by waiting for its first loads (s_waitcnt share of the wave time 24 -> 36 % against the plain kernel, which hides them
// behind its 440 set-up instructions).
template <int NB, bool PF>
__device__ __forceinline__ void winograd_persist_body(const WinoParams& p, float* smem) {
    constexpr int MO = 2, NT = 256, BN = 32 * NB, NW = 4;
    using ST = WinoStage<MO>;
    constexpr int W_NLD = ST::NLD, STAGE = ST::FLOATS, ROWF = ST::ROWE * W_CK;
    float* Rs = smem + W_ZERO;
    const int tid = threadIdx.x, lane = tid & 63, xi = __builtin_amdgcn_readfirstlane(tid >> 6);

    // block -> (part, group of the period, filter tile); consecutive ids of the sequence on one XCD (block b runs on XCD b % 8), the
    // filter tile fastest: the blocks that read the same staged pixels are neighbours in time and share an L2
    const int bid = blockIdx.x, nblk = gridDim.x;
    const int xcd = bid & 7, loc = bid >> 3, q8 = nblk >> 3, r8 = nblk & 7;
    const int t = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + loc;
    const int units = p.pg * p.n_tiles;
    const int part = fdiv(t, p.d_grp), unit = t - part * units;
    const int slot = fdiv(unit, p.d_ntiles), n_tile = unit - slot * p.n_tiles;
    const int it0 = part * p.chunk, it1 = min(it0 + p.chunk, p.periods);
    if (it0 >= it1) return;
    const int m0 = slot * W_TB, n0 = n_tile * BN;
    const int Lt = p.pg * W_TB;                            // tiles of one period (whole images): the geometry below is that of period 0

    // ---- staging offsets (see winograd_body: one slot per distinct pixel column of a tile row, chunk swizzle on the source side) ----
    // The geometry of the block — staging offsets, fragment addresses, store offsets: 16 words per thread — is parked in LDS behind the stage
    // buffers and read back at the top of every period (staging / fragment part) and before the write-out (store part): as plain loop invariants
    // these 13 values stay live across the MFMA loop AND the write-out, and the two-half kernel spilled 22 - 53 registers into the stage loop.
    static_assert(W_NLD <= 5, "geometry record layout");
    constexpr int PS_FLOATS = 4 * MO * W_TB * W_LDP;
    float* const Ps = PF ? smem + W_ZERO + 2 * STAGE : smem;                     // output-transform tiles [xi][b][tile][32 channels]
    wu32x4* Gs = reinterpret_cast<wu32x4*>(smem + W_ZERO + 2 * STAGE + (PF ? PS_FLOATS : 0));   // [3][256 threads] x 16 bytes: goff[0..3] | goff[4], ra[0..3] as 16-bit pairs | off[0..3]
    unsigned g0[8] = {WOOB_G, WOOB_G, WOOB_G, WOOB_G, WOOB_G, WOOB_G, WOOB_G, WOOB_G};
#pragma unroll
    for (int u = 0; u < W_NLD; ++u) {
        const int pz = (xi + NW * u) * 64 + lane;
        if (pz >= ST::ITEMS) continue;
        const int cpos = pz & 3, e = pz >> 2;
        const int i = e / ST::ROWE, re = e - i * ST::ROWE;
        const int r = re / W_NQ, q = re - r * W_NQ;
        if (r >= MO) continue;                             // the row's zero entry
        const int chunk = cpos ^ ((q >> 2) & 3);
        int m = m0 + q;
        int gr = fdiv(m, p.d_TW), tx = m - gr * p.TW;
        int xx = MO * tx + r - 1;
        if (m >= Lt || (tx == 0 && xx < 0)) {
            m -= 1;
            if (m < 0) continue;
            gr = fdiv(m, p.d_TW); tx = m - gr * p.TW;
            xx = MO * tx + MO + r - 1;
        }
        const int b = fdiv(gr, p.d_TH), ty = gr - b * p.TH;
        const int yy = MO * ty - 1 + i;
        if (m < Lt && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W)
            g0[u] = (unsigned)(((b * p.H + yy) * p.W + xx) * p.Cin + chunk * 4) << 2;
    }
    Gs[tid] = wu32x4{g0[0], g0[1], g0[2], g0[3]};
    // ---- fragment addressing ----
    const int h = lane >> 5;
    {
        int ra[4];
        const int tl = lane & 31;
        const int m = m0 + tl;                             // < Lt: a period holds whole groups
        const int gr = fdiv(m, p.d_TW), tx = m - gr * p.TW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int xx = MO * tx + j - 1;
            const int q = tl + j / MO;
            ra[j] = (unsigned)xx < (unsigned)p.W ? ((j % MO) * W_NQ + q) * W_CK + ((h ^ ((q >> 2) & 3)) << 2) : MO * W_NQ * W_CK + (h << 2);
        }
        Gs[NT + tid] = wu32x4{g0[4], ((unsigned)ra[0] & 0xFFFFu) | ((unsigned)ra[1] << 16), ((unsigned)ra[2] & 0xFFFFu) | ((unsigned)ra[3] << 16), 0u};
    }
    const int ia = xi == 0 ? 0 : (xi == 2 ? 2 : 1);
    const int ib = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float sgn = xi == 1 ? 1.f : -1.f;
    const f32x4 m1 = {p.neg_one, p.neg_one, p.neg_one, p.neg_one};
    const int roa = ia * ROWF, rob = ib * ROWF;
    // ---- filter fragments ----
    const int steps = p.stages * 2;
    const int hn = n_tile * NB;
    const int ut = hn / p.nhp, nh_g = hn - ut * p.nhp;
    const unsigned ubase = (unsigned)((((ut * steps) * 16 + 4 * xi) * p.nhp + nh_g) * 64 + lane) << 4;
    const unsigned ustep = 16u * p.nhp * 1024u, unu = p.nhp * 1024u;
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);
    // ---- write-out: per pass u the thread's (tile, output column, channel quad); store offsets of the first filter half (the second: + 128 bytes) ----
    constexpr int C4 = 32 / 4;
    const int c4 = tid % C4;
    {
        wu32x4 o4;
#pragma unroll
        for (int u = 0; u < MO; ++u) {
            const int rest = (tid + NT * u) / C4;
            const int tl = rest / MO, bq = rest - tl * MO;
            const int m = m0 + tl;
            const int b = fdiv(m, p.d_tpi), r = m - b * p.tpi;
            const int ty = fdiv(r, p.d_TW), tx = r - ty * p.TW;
            const int xx = MO * tx + bq;
#pragma unroll
            for (int a = 0; a < MO; ++a) {
                const int yy = MO * ty + a;
                o4[u * MO + a] = (xx < p.W && yy < p.H) ? (unsigned)(((b * p.H + yy) * p.W + xx) * p.Cout + n0 + c4 * 4) << 2 : WOOB;
            }
        }
        Gs[2 * NT + tid] = o4;
    }
    const f32x4 one = {1.f, 1.f, 1.f, 1.f}, nul = {0.f, 0.f, 0.f, 0.f};
    const float lo = p.relu ? 0.f : -INFINITY;
    bool nvh[NB];
#pragma unroll
    for (int hh = 0; hh < NB; ++hh) nvh[hh] = n0 + 32 * hh + c4 * 4 < p.Cout;

    auto mfma_group = [&](f32x16 (&ac)[4], const f32x4 (&v)[4], const f32x4 (&uu)[4]) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
                ac[nu] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[nu][tt], uu[nu][tt], ac[nu], 0, 0, 0);
    };
    auto u_load_h = [&](f32x4 (&dst)[4], int step_, int half) {
        const int step = min(step_, steps - 1);            // (past the last step: the last step's fragments again, no per-lane select)
        const unsigned o = ubase + (unsigned)half * 1024u + (unsigned)step * ustep;
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) dst[nu] = wbuf_load4(ur, o + nu * unu);
    };

    // requests of a period's first two stages + first filter fragments (PF: issued during the write-out of the period before)
    f32x4 ua[4], ub[4];
    auto first_requests = [&](int it) {
        const __amdgpu_buffer_rsrc_t xq = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (long long)it * p.x_period_floats), 0, p.x_bytes, 0x00020000);
        u_load_h(ua, 0, 0);
        int tid_p = tid;
        asm volatile("" : "+v"(tid_p));
        const wu32x4 ga = Gs[tid_p], gb = Gs[NT + tid_p];
        unsigned goff[5];
#pragma unroll
        for (int k = 0; k < 4; ++k) goff[k] = ga[k];
        goff[4] = gb[0];
#pragma unroll
        for (int st = 0; st < 2; ++st)
            if (st < p.stages) {
#pragma unroll
                for (int u = 0; u < W_NLD; ++u)
                    if (xi + NW * u < ST::NDMA)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(xq, (wlds_void*)(Rs + st * STAGE + (xi + NW * u) * 256), 16,
                                                                 goff[u] + (unsigned)st * (W_CK * 4), 0, 0, 0);
            }
    };
    if constexpr (PF) {
        if (tid < W_ZERO / 4) *reinterpret_cast<f32x4*>(&smem[tid * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};
        first_requests(it0);
    }

    for (int it = it0; it < it1; ++it) {
        // the period's tensors: base pointers advance, offsets stay (32-bit offsets inside a period-aligned window of the tensor)
        const float* xb = p.x + (long long)it * p.x_period_floats;
        float* yb = p.y + (long long)it * p.y_period_floats;
        const float* rb = p.res ? p.res + (long long)it * p.y_period_floats : nullptr;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, p.x_bytes, 0x00020000);
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));                    // (opaque: the reads below are per period, not hoisted)
        int ra[4];
        {
            const wu32x4 gc = Gs[NT + tid_o];
            ra[0] = (int)(short)(gc[1] & 0xFFFFu); ra[1] = (int)(short)(gc[1] >> 16);
            ra[2] = (int)(short)(gc[2] & 0xFFFFu); ra[3] = (int)(short)(gc[2] >> 16);
        }
        auto make_v = [&](f32x4 (&v)[4], const float* Rb, int x8) {
            f32x4 tc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int a = ra[j];
                const f32x4 da = *reinterpret_cast<const f32x4*>(Rb + roa + (a ^ x8));
                const f32x4 db = *reinterpret_cast<const f32x4*>(Rb + rob + (a ^ x8));
                tc[j] = da + sgn * db;
            }
            v[0] = tc[0] + m1 * tc[2]; v[1] = tc[1] + tc[2]; v[2] = tc[2] + m1 * tc[1]; v[3] = tc[1] + m1 * tc[3];
        };
        auto stage_dma = [&](int buf, int st) {            // (the staging offsets are read from the record at every use: 5 fewer registers live across the stage loop)
            int tid_p = tid_o;
            asm volatile("" : "+v"(tid_p));
            const wu32x4 ga = Gs[tid_p], gb = Gs[NT + tid_p];
            unsigned goff[5];
#pragma unroll
            for (int k = 0; k < 4; ++k) goff[k] = ga[k];
            goff[4] = gb[0];
#pragma unroll
            for (int u = 0; u < W_NLD; ++u)
                if (xi + NW * u < ST::NDMA)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (wlds_void*)(Rs + buf * STAGE + (xi + NW * u) * 256), 16,
                                                             goff[u] + (unsigned)st * (W_CK * 4), 0, 0, 0);
        };
        if constexpr (!PF) {
            u_load_h(ua, 0, 0);
            stage_dma(0, 0);
            if (p.stages > 1) stage_dma(1, 1);
            if (tid < W_ZERO / 4) *reinterpret_cast<f32x4*>(&smem[tid * 4]) = f32x4{0.f, 0.f, 0.f, 0.f};   // (the output-transform tiles of the period before lay over the zero pixel)
        }
        f32x16 accs[NB][4];
#pragma unroll
        for (int hh = 0; hh < NB; ++hh)
#pragma unroll
            for (int nu = 0; nu < 4; ++nu)
#pragma unroll
                for (int e = 0; e < 16; ++e) accs[hh][nu][e] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the period's first two stages have landed
        lds_barrier();

        if constexpr (NB == 2) {
            for (int st = 0; st < p.stages; ++st) {
                const int buf = st & 1;
                const float* Rb = Rs + buf * STAGE;
                f32x4 v[4], v1[4];
                u_load_h(ub, 2 * st, 1);
                __builtin_amdgcn_sched_barrier(0);
                make_v(v, Rb, 0);
                mfma_group(accs[0], v, ua);
                make_v(v1, Rb, 8);
                __builtin_amdgcn_sched_barrier(0);
                u_load_h(ua, 2 * st + 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(accs[NB - 1], v, ub);
                __builtin_amdgcn_sched_barrier(0);
                u_load_h(ub, 2 * st + 1, 1);
                if (st + 1 < p.stages && st > 0) stage_dma(buf ^ 1, st + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nu = 0; nu < 4; ++nu) v[nu] = v1[nu];
                mfma_group(accs[0], v, ua);
                __builtin_amdgcn_sched_barrier(0);
                u_load_h(ua, 2 * st + 2, 0);
                __builtin_amdgcn_sched_barrier(0);
                mfma_group(accs[NB - 1], v, ub);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        } else {
            for (int st = 0; st < p.stages; ++st) {
                const int buf = st & 1;
                const float* Rb = Rs + buf * STAGE;
                f32x4 v[4];
                u_load_h(ub, 2 * st + 1, 0);
                if (st + 1 < p.stages && st > 0) stage_dma(buf ^ 1, st + 1);
                __builtin_amdgcn_sched_barrier(0);
                make_v(v, Rb, 0);
                mfma_group(accs[0], v, ua);
                __builtin_amdgcn_sched_barrier(0);
                u_load_h(ua, 2 * st + 2, 0);
                __builtin_amdgcn_sched_barrier(0);
                make_v(v, Rb, 8);
                mfma_group(accs[0], v, ub);
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        }

        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(yb, 0, p.y_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rb), 0, rb ? p.y_bytes : 0u, 0x00020000);
        // (the write-out's LDS addresses are formed from the opaque copy of the thread id too: as loop invariants hipcc would keep all ~40 of
        // them in registers across the MFMA loop)
        const int lane_o = tid_o & 63, c4o = tid_o % C4;
        const wu32x4 o4 = Gs[2 * NT + tid_o];
        unsigned off[MO][MO];
#pragma unroll
        for (int u = 0; u < MO; ++u)
#pragma unroll
            for (int a = 0; a < MO; ++a) off[u][a] = o4[u * MO + a];
#pragma unroll
        for (int hh = 0; hh < NB; ++hh) {
            f32x16 (&acch)[4] = accs[hh];
            const unsigned hoff = (unsigned)hh * 128u;
            const int nq = n0 + 32 * hh + c4o * 4;
            const f32x4 sc = (nvh[hh] && p.scale) ? *reinterpret_cast<const f32x4*>(p.scale + nq) : one;
            const f32x4 bi = (nvh[hh] && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + nq) : nul;
            f32x4 rs[MO][MO];
#pragma unroll
            for (int u = 0; u < MO; ++u)
#pragma unroll
                for (int a = 0; a < MO; ++a)
                    rs[u][a] = rb ? wbuf_load4(rr, (nvh[hh] && off[u][a] != WOOB) ? off[u][a] + hoff : WOOB) : nul;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (PF) {                            // the stage buffers are free (barrier after the last stage), this half's skip-connection values are requested
                if (hh == 0 && it + 1 < it1) first_requests(it + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                const int cl = lane_o & 31;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane_o >> 5);
                    const float m0v = acch[0][e], m1v = acch[1][e], m2v = acch[2][e], m3v = acch[3][e];
                    Ps[((xi * MO + 0) * W_TB + row) * W_LDP + cl] = m0v + m1v + m2v;
                    Ps[((xi * MO + 1) * W_TB + row) * W_LDP + cl] = m1v - m2v - m3v;
                }
            }
            lds_barrier();                                 // (LDS hand-off only: skip-connection loads, the next period's requests and the stores of the half before stay in flight)
#pragma unroll
            for (int u = 0; u < MO; ++u) {
                const int rest = (tid_o + NT * u) / C4;
                const int tl = rest / MO, bq = rest - tl * MO;
                f32x4 pq[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) pq[k] = *reinterpret_cast<const f32x4*>(&Ps[((k * MO + bq) * W_TB + tl) * W_LDP + c4o * 4]);
                f32x4 yv[MO];
                yv[0] = pq[0] + pq[1] + pq[2];
                yv[1] = pq[1] - pq[2] - pq[3];
#pragma unroll
                for (int a = 0; a < MO; ++a) {
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = fmaxf(yv[a][c] * sc[c] + bi[c] + rs[u][a][c], lo);
                    wbuf_store4(yr, (nvh[hh] && off[u][a] != WOOB) ? off[u][a] + hoff : WOOB, o);
                }
            }
            lds_barrier();                                 // the next half's / the next period's stage goes where these tiles were read (the stores stay in flight)
        }
    }
}

template <int NB, bool PF>
__global__ __launch_bounds__(256, (NB == 2 || PF) ? 2 : 3) void winograd_persist_kernel(WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    winograd_persist_body<NB, PF>(p, smem);
}

// GATHER: the data gradient of the transposed conv (reduction over the four pixel phases of dz x channels); a separate instantiation —
// its phase changes keep the loader's geometry live across the stage loop, which costs the other variants 37 spilled registers
template <int MO, bool BNB, bool GATHER, int NB = 1>
__global__ __launch_bounds__(256, NB == 2 ? 2 : 3) void winograd_kernel(WinoParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if constexpr (GATHER) {
        // Written as a loop over the tiles of this block — which runs exactly once, the grid has one block per tile — with the parameter
        // block read through an opaque pointer to the kernel-argument segment (WinoParams is the only argument) and an opaque thread id.
        // In this form hipcc 7.2 splits the stage loop of the gather mode by input phase: a 409-instruction inner loop without spill
        // traffic, the re-loads (34 scratch loads, 80 lane reads per trip in the plain form) only at the four phase changes.  deconv3's
        // data gradient at 120 crops 1.34 -> 1.11 ms, deconv2's 0.71 -> 0.61, deconv1's 0.37 -> 0.31 (same box, alternating).  The other
        // instantiations are 2 - 3 % slower in this form and stay plain.
        typedef const __attribute__((address_space(4))) WinoParams* KArg;
        const int total = gridDim.x;
        for (int bid = blockIdx.x; bid < total; bid += gridDim.x) {
            KArg pp = (KArg)__builtin_amdgcn_kernarg_segment_ptr();
            asm volatile("" : "+s"(pp));
            winograd_body<MO, BNB, GATHER, NB>(*(const WinoParams*)pp, smem, bid, total);
            __syncthreads();
        }
    } else {
        winograd_body<MO, BNB, GATHER, NB>(p, smem, blockIdx.x, gridDim.x);
    }
}

// U = G g G^T (winograd_pack.h): one block per (32 output channels, 8 input channels) (x 4 phases for the transposed conv)
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int NH, int mode, int w_i) {
    wino_pack_block(w, out, mode, w_i, Cout, Cin, NH, blockIdx.x, threadIdx.x);
}

static std::atomic<unsigned> g_wino_lds_done[12];  // (MO - 2) * 2 + BNB, + 4 for the gather instantiations, + 6 for the two-half blocks
static std::atomic<int> g_wino_halves{2};          // vatl_tune_set(21, v): filter halves per block where the layer allows two (1 = always one)
int wino_set_halves(int v) { g_wino_halves.store(v, std::memory_order_relaxed); return 0; }
static std::atomic<int> g_wino_ablate{0};
static std::atomic<int> g_wino_group_kb{2048};     // vatl_tune_set(18, v): KB of filter slices per group of the tile order (0 = one slice)
int wino_set_group_kb(int v) { g_wino_group_kb.store(v, std::memory_order_relaxed); return 0; }
int wino_set_ablate(int bits) { g_wino_ablate.store(bits, std::memory_order_relaxed); return 0; }

static std::atomic<int> g_wino_persist{8};         // vatl_tune_set(22, v): 3x3 layers with at most v 16-channel stages take the persistent route (0 = never)
int wino_set_persist(int v) { g_wino_persist.store(v, std::memory_order_relaxed); return 0; }
static std::atomic<unsigned> g_wino_persist_lds_done[4];
static std::atomic<int> g_wino_persist_pf{2};      // vatl_tune_set(24, bits): 1 = prefetching variant for one-half blocks (two blocks per CU instead of three), 2 = for two-half blocks
int wino_set_persist_pf(int v) { g_wino_persist_pf.store(v, std::memory_order_relaxed); return 0; }

constexpr int kWinoMaxLds = 64 * 1024;            // upper bound of a block's dynamic LDS (ns <= 128: two 32 KB stages)

template <int MO, bool BNB, bool GATHER = false, int NB = 1>
static int launch_wino(const WinoParams& p, int phases, hipStream_t st) {
    auto kern = winograd_kernel<MO, BNB, GATHER, NB>;
    if (int rc = ensure_dynamic_lds((const void*)kern, kWinoMaxLds, g_wino_lds_done[(NB == 2 ? 6 : 0) + (GATHER ? 4 + (BNB ? 1 : 0) : (MO - 2) * 2 + (BNB ? 1 : 0))], "winograd")) return rc;
    const int loop = W_ZERO + 2 * WinoStage<MO>::FLOATS, epi = 4 * MO * W_TB * W_LDP, sta = 2 * 256 * 4;
    const int smem = std::max(loop, std::max(epi, sta)) * (int)sizeof(float);
    if (smem > kWinoMaxLds) return fail(VATL_EINVAL, "winograd: %d bytes of LDS per block", smem);
    hipLaunchKernelGGL(kern, dim3(p.m_tiles * p.n_tiles * phases), dim3(256), smem, st, p);
    meter_add(1, 2.0 * ((double)p.m_tiles * W_TB) * ((double)p.n_tiles * 32 * NB) * 16.0 * ((double)p.stages * W_CK) * phases);
    meter_route(BNB ? kRouteWinoBnBwd : (NB == 2 ? kRouteWino2H : kRouteWino));
    return check_launch("winograd");
}

static thread_local int tl_wino_route = 0;         // what the calling thread's last F(2x2,3x3) forward launch took (vatl_winograd_last_route)

static long long gcd_ll(long long a, long long b) { while (b) { const long long t = a % b; a = b; b = t; } return a; }

// The persistent route of a forward F(2x2,3x3) launch: images [0, periods * Pi) in ONE launch of blocks that each keep a (group, filter tile) of
// the period; returns the number of images it covered (0: not taken, the caller runs the plain kernel on everything; otherwise the caller runs
// the plain kernel on the remaining N % Pi images).  Taken when the blocks are short (<= g_wino_persist stages) and a grid of whole rounds of
// resident blocks can be cut with >= 90 % of its slots busy and >= 2 periods per block.
static int launch_wino_persist(const WinoParams& base, int N, bool two, hipStream_t st, int* covered) {
    *covered = 0;
    const int maxst = g_wino_persist.load(std::memory_order_relaxed);
    if (maxst <= 0 || base.stages > maxst || base.deconv || base.gather || base.stats || base.bz) return 0;
    const long long Lt = (long long)base.tpi / gcd_ll(base.tpi, W_TB) * W_TB;          // lcm(tiles per image, 32)
    const int Pi = (int)(Lt / base.tpi), Pg = (int)(Lt / W_TB);
    const int periods = N / Pi;
    if (periods < 8 || Pg > 4096) return 0;
    const int NBh = two ? 2 : 1;
    const int n_tiles = cdiv(base.Cout, 32 * NBh);
    const int units = Pg * n_tiles, slots = (two || (g_wino_persist_pf.load(std::memory_order_relaxed) & 1)) ? 512 : 768;
    if (units > slots) return 0;
    // whole rounds of resident blocks: k rounds -> at most k * slots / units parts; keep the cut with the best slot occupancy
    int best_chunk = 0, best_parts = 0;
    double best_eff = 0.0;
    for (int k = 1; k <= 4; ++k) {
        const int np_max = std::min(periods, k * slots / units);
        if (np_max < 1) continue;
        const int chunk = cdiv(periods, np_max), parts = cdiv(periods, chunk);
        if (chunk < 2) break;
        const int rounds = cdiv((long long)parts * units, slots);
        const double eff = (double)periods * units / ((double)rounds * slots * chunk);
        if (eff > best_eff + 0.02) { best_eff = eff; best_chunk = chunk; best_parts = parts; }
    }
    // measured (tools/exp_persist.py, profiles/r04_notes.md): two- to four-stage blocks gain 4 - 7 % wherever the cut keeps the slots busy; five to eight
    // stages gain 3 - 5 % with long walks and an even cut, and lose 3 - 4 % with two periods per block (hr.b128) or an uneven cut (r152.l2.c2)
    if (best_eff < 0.9 || (base.stages > 4 && (best_eff < 0.95 || best_chunk < 8))) return 0;
    WinoParams p = base;
    p.N = Pi; p.Mtiles = (int)Lt;
    p.pg = Pg; p.periods = periods; p.chunk = best_chunk;
    p.n_tiles = n_tiles; p.m_tiles = Pg;
    p.x_period_floats = (long long)Pi * base.H * base.W * base.Cin;
    p.y_period_floats = (long long)Pi * base.H * base.W * base.Cout;
    p.x_bytes = (unsigned)(p.x_period_floats * 4); p.y_bytes = (unsigned)(p.y_period_floats * 4);
    p.d_grp = make_fastdiv((unsigned)units); p.d_ntiles = make_fastdiv((unsigned)n_tiles);
    const int loop = W_ZERO + 2 * WinoStage<2>::FLOATS, epi = 4 * 2 * W_TB * W_LDP;
    const int pfbits = g_wino_persist_pf.load(std::memory_order_relaxed);
    const bool pf = two ? (pfbits & 2) != 0 : (pfbits & 1) != 0;
    // geometry records of the 256 threads behind the stage buffers (PF: behind the output-transform tiles, which then have a region of their own)
    const int smem = ((pf ? loop + epi : std::max(loop, epi)) + 3 * 256 * 4) * (int)sizeof(float);
    const unsigned grid = (unsigned)(best_parts * units);
    auto go = [&](auto kern, int slot) -> int {
        if (int rc = ensure_dynamic_lds((const void*)kern, 96 * 1024, g_wino_persist_lds_done[slot], "winograd_persist")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, st, p);
        return 0;
    };
    if (int rc = two ? (pf ? go(winograd_persist_kernel<2, true>, 3) : go(winograd_persist_kernel<2, false>, 1))
                     : (pf ? go(winograd_persist_kernel<1, true>, 2) : go(winograd_persist_kernel<1, false>, 0))) return rc;
    meter_add(1, 2.0 * ((double)periods * Lt) * ((double)n_tiles * 32 * NBh) * 16.0 * ((double)base.stages * W_CK));
    meter_route(kRouteWinoPersist);
    *covered = periods * Pi;
    return check_launch("winograd_persist");
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

// ConvTranspose2d(4, 2, 1) filter (Cin, Cout, 4, 4) -> four phase filters G g_phase G^T, [phase][fragment order of vatl_pack_winograd_weight]
extern "C" int64_t vatl_winograd_deconv_weight_floats(int Cout, int Cin) { return 4 * vatl_winograd_weight_floats(Cout, Cin); }

extern "C" int vatl_pack_winograd_deconv_weight(const float* w, float* u, int Cout, int Cin, void* stream) {
    if (!w || !u || Cout <= 0 || Cin <= 0) return fail(VATL_EINVAL, "pack_winograd_deconv_weight: null pointer or empty filter");
    if (Cin % 16 != 0) return fail(VATL_EINVAL, "pack_winograd_deconv_weight: Cin %d must be a multiple of 16", Cin);
    const int pad = vatl_winograd_cout_pad(Cout);
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)(4 * (pad / 32) * (Cin / 8))), dim3(256), 0, (hipStream_t)stream, w, u, Cout, Cin, wino_nh(Cout),
                       2, Cout);
    return check_launch("wino_pack_deconv");
}

struct WinoBn { const float *z, *mask_y, *scale, *bias, *mean, *invstd; };

// MO = 2: 3x3 / stride 1 / pad 1 conv (y: N x H x W x Cout);  MO = 3: ConvTranspose2d(4, 2, 1) (y: N x 2H x 2W x Cout), or with
// `gather` its data gradient: x = dz (N x 2H x 2W x Cin), y = dx (N x H x W x Cout), the four input phases part of the reduction
static int winograd_impl(int MO, const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y,
                         double* stats, int64_t* row_blocks_used, int N, int H, int W, int Cin, int Cout, int relu, void* stream,
                         const WinoBn* fuse = nullptr, bool gather = false) {
    if (!x || !u || !y || N <= 0 || H <= 0 || W <= 0) return fail(VATL_EINVAL, "winograd: null pointer or empty batch");
    if (Cin % 16 != 0 || (Cout & 3)) return fail(VATL_EINVAL, "winograd: Cin %d must be a multiple of 16 and Cout %d of 4", Cin, Cout);
    WinoParams p{};
    p.neg_one = -1.0f;
    p.x = x; p.u = u; p.scale = scale; p.bias = bias; p.res = residual; p.y = y; p.stats = stats;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
    p.TH = (H + MO - 1) / MO; p.TW = (W + MO - 1) / MO; p.tpi = p.TH * p.TW;
    const bool deconv = MO == 3 && !gather;
    const int phases = deconv ? 4 : 1, os = deconv ? 2 : 1;
    const long long mt = (long long)N * p.tpi, xe = (long long)N * H * W * Cin * (gather ? 4 : 1), ye = (long long)N * H * W * os * os * Cout;
    const long long ue = vatl_winograd_weight_floats(Cout, Cin);
    if (xe > (long long)(WOOB_G / 4) || ye >= (1LL << 30) || ue >= (1LL << 28) || mt >= (1LL << 30) || Cin / W_CK >= 1024)
        return fail(VATL_EINVAL, "winograd: a tensor exceeds 2^30 elements (32-bit buffer offsets); split the batch");
    p.Mtiles = (int)mt;
    p.nhp = wino_nh(Cout);
    p.spp = Cin / W_CK;
    p.stages = gather ? 4 * p.spp : p.spp;
    p.pad_y = 1; p.pad_x = 1; p.OH = H * os; p.OW = W * os; p.os = os; p.ooy = 0; p.oox = 0;
    p.deconv = deconv; p.gather = gather; p.u_phase_floats = ue;
    p.x_bytes = (unsigned)(xe * 4); p.y_bytes = (unsigned)(ye * 4); p.u_bytes = (unsigned)(ue * 4 * (gather ? 4 : 1));
    p.ablate = g_wino_ablate.load(std::memory_order_relaxed);
    // Two filter halves per block (one staged input and one input transform for 64 output channels, 128 accumulator registers, two blocks
    // per CU) where the padded filter has an even number of halves (packed in pairs: nhp = 2) and the launch still has enough blocks for the
    // 512 slots that leaves: at 1024 crops every layer gains 5 - 12 %; small launches lose (120 crops: l4.c2 360 blocks 134 -> 154 us,
    // deconv1 368 blocks 494 -> 578; 32 crops: l2.c2 384 blocks 42 -> 47) where 432 blocks and more win (R152 l3.c2 at 32 crops 96 -> 86 us,
    // l3.c2 at 120 crops, 720 blocks, 151 -> 134).  The gather instantiations too (256 VGPRs, 13 spilled against 27 - 31 in the small shape: deconv3's
    // data gradient at 120 crops 1.11 -> 1.04 ms, deconv1's 0.61 -> 0.56).
    p.m_tiles = cdiv(mt, W_TB);
    const int hv = g_wino_halves.load(std::memory_order_relaxed);          // 1: never, 2: with the block-count floor, 3: wherever possible
    const bool two = p.nhp == 2 && vatl_winograd_cout_pad(Cout) % 64 == 0 && hv >= 2 &&
                     (hv == 3 || (long long)p.m_tiles * cdiv(Cout, 64) * phases >= 400);
    const int NBh = two ? 2 : 1;
    p.n_tiles = cdiv(Cout, 32 * NBh);
    hipStream_t st = (hipStream_t)stream;
    if (MO == 2 && !fuse && !stats && !gather) {
        // short blocks: the persistent route (geometry computed once per block, periods of whole images walked by base address); the images
        // that do not fill a period go through the plain kernel below
        p.d_TH = make_fastdiv(p.TH); p.d_TW = make_fastdiv(p.TW); p.d_tpi = make_fastdiv(p.tpi);
        int covered = 0;
        if (int rc = launch_wino_persist(p, N, two, st, &covered)) return rc;
        tl_wino_route = covered == 0 ? 0 : (covered == N ? 1 : 2);
        if (covered == N) return 0;
        if (covered > 0) {
            const int rc = winograd_impl(MO, x + (long long)covered * H * W * Cin, u, scale, bias, residual ? residual + (long long)covered * H * W * Cout : nullptr,
                                 y + (long long)covered * H * W * Cout, nullptr, nullptr, N - covered, H, W, Cin, Cout, relu, stream);
            tl_wino_route = 2;
            return rc;
        }
    }
    if (row_blocks_used) *row_blocks_used = (int64_t)p.m_tiles * phases;
    // a slice is Cin * 2 KB (x 4 input phases in the gather mode); at least two per group (deconv1, Cin = 2048: 4 MB slices, 4326 -> 4135 us
    // with two), unless the knob says 0
    const int gkb = g_wino_group_kb.load(std::memory_order_relaxed);
    p.rn = std::max(1, std::min(p.n_tiles * phases, std::max(gkb > 0 ? 2 : 1, gkb / (2 * Cin * NBh * (gather ? 4 : 1)))));
    const int units = p.n_tiles * phases;
    p.d_TH = make_fastdiv(p.TH); p.d_TW = make_fastdiv(p.TW); p.d_tpi = make_fastdiv(p.tpi);
    p.d_grp = make_fastdiv((unsigned)(p.m_tiles * p.rn)); p.d_rn = make_fastdiv(p.rn); p.d_ntiles = make_fastdiv(p.n_tiles);
    p.d_rn_last = make_fastdiv(units % p.rn ? units % p.rn : p.rn);
    if ((long long)p.m_tiles * units >= (1LL << 31)) return fail(VATL_EINVAL, "winograd: too many blocks");
    if (fuse) {
        if (deconv) return fail(VATL_EINVAL, "winograd: the BatchNorm-backward epilogue exists for the data-gradient launches only");
        p.bz = fuse->z; p.bmy = fuse->mask_y; p.bsc = fuse->scale; p.bbi = fuse->bias; p.bmu = fuse->mean; p.bis = fuse->invstd;
        if (gather) return two ? launch_wino<3, true, true, 2>(p, phases, st) : launch_wino<3, true, true>(p, phases, st);
        if (two) return MO == 2 ? launch_wino<2, true, false, 2>(p, phases, st) : launch_wino<3, true, false, 2>(p, phases, st);
        return MO == 2 ? launch_wino<2, true>(p, phases, st) : launch_wino<3, true>(p, phases, st);
    }
    if (gather) return two ? launch_wino<3, false, true, 2>(p, phases, st) : launch_wino<3, false, true>(p, phases, st);
    if (two) return MO == 2 ? launch_wino<2, false, false, 2>(p, phases, st) : launch_wino<3, false, false, 2>(p, phases, st);
    return MO == 2 ? launch_wino<2, false>(p, phases, st) : launch_wino<3, false>(p, phases, st);
}

extern "C" int vatl_conv3x3_winograd_fwd(const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y,
                                         int N, int H, int W, int Cin, int Cout, int relu, void* stream) {
    return winograd_impl(2, x, u, scale, bias, residual, y, nullptr, nullptr, N, H, W, Cin, Cout, relu, stream);
}

extern "C" int vatl_winograd_last_route(void) { return tl_wino_route; }

extern "C" int64_t vatl_winograd_stats_row_blocks(int64_t N, int H, int W) { return (N * ((H + 1) / 2) * ((W + 1) / 2) + W_TB - 1) / W_TB; }

extern "C" int vatl_conv3x3_winograd_fwd_stats(const float* x, const float* u, float* y, double* stats, int64_t* row_blocks_used, int N, int H,
                                               int W, int Cin, int Cout, void* stream) {
    if (!stats || !row_blocks_used) return fail(VATL_EINVAL, "conv3x3_winograd_fwd_stats: null statistics buffer");
    return winograd_impl(2, x, u, nullptr, nullptr, nullptr, y, stats, row_blocks_used, N, H, W, Cin, Cout, 0, stream);
}

// Data-gradient launch fused with the reduction pass of the consumer layer's BatchNorm backward: the Winograd counterpart of
// vatl_conv2d_fwd_ex_bnbwd (same masks, same statistics layout; u = the data-gradient packing of the filter).
extern "C" int vatl_conv3x3_winograd_fwd_bnbwd(const float* x, const float* u, const float* residual, float* y, int N, int H, int W, int Cin, int Cout,
                                               const float* bn_z, const float* bn_mask_y, const float* bn_scale, const float* bn_bias,
                                               const float* bn_mean, const float* bn_invstd, double* stats, int64_t* row_blocks_used, void* stream) {
    if (!bn_z || !stats || !row_blocks_used || !bn_mean || !bn_invstd || (bn_scale && !bn_bias))
        return fail(VATL_EINVAL, "conv3x3_winograd_fwd_bnbwd: needs z, mean, invstd and a statistics buffer");
    const WinoBn bn{bn_z, bn_mask_y, bn_scale, bn_bias, bn_mean, bn_invstd};
    return winograd_impl(2, x, u, nullptr, nullptr, residual, y, stats, row_blocks_used, N, H, W, Cin, Cout, 0, stream, &bn);
}

// ConvTranspose2d(4, 2, 1) + folded BatchNorm + ReLU (inference) / + batch statistics (training forward) on the F(3x3, 2x2) route
extern "C" int vatl_deconv4x4s2_winograd_fwd(const float* x, const float* u, const float* scale, const float* bias, float* y, int N, int H, int W,
                                             int Cin, int Cout, int relu, void* stream) {
    return winograd_impl(3, x, u, scale, bias, nullptr, y, nullptr, nullptr, N, H, W, Cin, Cout, relu, stream);
}

extern "C" int64_t vatl_winograd_deconv_stats_row_blocks(int64_t N, int H, int W) { return 4 * ((N * ((H + 2) / 3) * ((W + 2) / 3) + W_TB - 1) / W_TB); }

extern "C" int vatl_deconv4x4s2_winograd_fwd_stats(const float* x, const float* u, float* y, double* stats, int64_t* row_blocks_used, int N, int H,
                                                   int W, int Cin, int Cout, void* stream) {
    if (!stats || !row_blocks_used) return fail(VATL_EINVAL, "deconv4x4s2_winograd_fwd_stats: null statistics buffer");
    return winograd_impl(3, x, u, nullptr, nullptr, nullptr, y, stats, row_blocks_used, N, H, W, Cin, Cout, 0, stream);
}

// Data gradient of ConvTranspose2d(4, 2, 1): dx[ci][y][x] = sum_{ky,kx,co} dz[co][2y - 1 + ky][2x - 1 + kx] W[ci][co][ky][kx] — a 4x4 / stride 2 conv,
// i.e. the sum over the four pixel phases of dz of 2x2 convolutions: F(3x3, 2x2) with the reduction over (phase, channel).  w = the layer's
// (Cin, Cout, 4, 4) weight; u receives 4 x vatl_winograd_weight_floats(Cin, Cout) floats.  dz (N, 2H, 2W, Cout) -> dx (N, H, W, Cin) (+ residual).
extern "C" int64_t vatl_winograd_deconv_dgrad_weight_floats(int Cin, int Cout) { return 4 * vatl_winograd_weight_floats(Cin, Cout); }

extern "C" int vatl_pack_winograd_deconv_dgrad_weight(const float* w, float* u, int Cin, int Cout, void* stream) {
    if (!w || !u || Cout <= 0 || Cin <= 0) return fail(VATL_EINVAL, "pack_winograd_deconv_dgrad_weight: null pointer or empty filter");
    if (Cout % 16 != 0) return fail(VATL_EINVAL, "pack_winograd_deconv_dgrad_weight: Cout %d must be a multiple of 16", Cout);
    const int pad = vatl_winograd_cout_pad(Cin);          // the packed filter's output channels are the layer's Cin, its reduction channels the layer's Cout
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)(4 * (pad / 32) * (Cout / 8))), dim3(256), 0, (hipStream_t)stream, w, u, Cin, Cout, wino_nh(Cin),
                       3, Cout);
    return check_launch("wino_pack_deconv_dgrad");
}

extern "C" int vatl_deconv4x4s2_winograd_dgrad(const float* dz, const float* u, const float* residual, float* dx, int N, int H, int W, int Cin,
                                               int Cout, void* stream) {
    return winograd_impl(3, dz, u, nullptr, nullptr, residual, dx, nullptr, nullptr, N, H, W, Cout, Cin, 0, stream, nullptr, true);
}

extern "C" int vatl_deconv4x4s2_winograd_dgrad_bnbwd(const float* dz, const float* u, const float* residual, float* dx, int N, int H, int W, int Cin,
                                                     int Cout, const float* bn_z, const float* bn_mask_y, const float* bn_scale,
                                                     const float* bn_bias, const float* bn_mean, const float* bn_invstd, double* stats,
                                                     int64_t* row_blocks_used, void* stream) {
    if (!bn_z || !stats || !row_blocks_used || !bn_mean || !bn_invstd || (bn_scale && !bn_bias))
        return fail(VATL_EINVAL, "deconv4x4s2_winograd_dgrad_bnbwd: needs z, mean, invstd and a statistics buffer");
    const WinoBn bn{bn_z, bn_mask_y, bn_scale, bn_bias, bn_mean, bn_invstd};
    return winograd_impl(3, dz, u, nullptr, nullptr, residual, dx, stats, row_blocks_used, N, H, W, Cout, Cin, 0, stream, &bn, true);
}
