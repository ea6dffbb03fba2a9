// Winograd F(4x4, 3x3) for the 3x3 / stride 1 / pad 1 layers whose output grid is a whole number of 4x4 tiles and whose reduction is long enough
// to amortise the larger transforms (ResNet stages 2 / 3: 128 / 256 channels at 32x24 / 16x12; HRNet's 128-channel branch): a 4x4 output tile from
// its 6x6 input tile with 36 multiplies per (input channel, output channel) where F(2x2, 3x3) needs 64 and the direct sum 144 — 0.5625x the MFMAs
// of csrc/conv_winograd.hip.  Interpolation points (0, 1, -1, 2, -2, inf):
//
//     Y = A^T [ (G g G^T) .* (B^T d B) ] A        B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//                                                 A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//                                                 G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//
// still exact fp32 products with fp32 accumulation; U = G g G^T is formed in float64 by the packer.  Accuracy: on the reference's golden crops the
// heat-maps of SimplePose-R50 with these eight layers on F(4x4) are 1.35e-6 (relative, max) from the float64 network where torch's fp32 network is
// 1.29e-6 (tools/probes/f4_accuracy.py, CPU); the tests hold the kernel to the F(2x2) route's tolerance and the stream route to the golden arg-max.
//
// Block = 16 tiles (flat index over image, tile row, tile column) x 64 output channels x all 36 positions, four waves, v_mfma_f32_16x16x4_f32:
//   * wave w owns the nine positions 9 w .. 9 w + 8 of the row-major 6 x 6 domain = one and a half rows, so it forms two row combinations
//     (B^T along the rows: 3 instructions per column and channel) and nine column combinations per (tile, channel): 2.3 vector instructions per
//     MFMA (vector instructions and fp32 MFMAs add up on a SIMD: profiles/r04_notes.md).  144 accumulator registers, two blocks per CU.
//   * A operand: lane (tile = lane % 16, quad = lane / 16) reads ITS tile's pixels, four consecutive channels per 16-byte read = the four MFMA k-steps
//     (k-step s multiplies channels {4 quad + s}: the same permutation on both operands).  The raw pixels of the block's 16 tiles x 36 pixels x 16
//     channels are staged by LDS-DMA (double buffer, one stage ahead; out-of-image pieces are requested out of range = the zero padding), one 1 KB
//     request per patch pixel: lane = (tile, channel quad), the quad rotated by 2 (tile / 8) so that the 16 lanes of a ds_read_b128 service group fall on
//     16 different bank slots.  A lane's requests differ by wave-uniform pixel offsets only: one base register + a 12-bit inside-the-image mask.
//   * B operand (U): packed in fragment order [stage][position][16-channel column block][lane][k-step]: a wave's read is 1 KB contiguous, the four column
//     blocks of a position 4 KB; straight from L2 into registers one position ahead.  A block needs the 64-channel slice of U (36 x Cin x 256 bytes:
//     2.4 MB at 256 channels); the block -> (tile group, slice) map gives every XCD ONE slice, so that it stays in that XCD's L2.
//   * Output transform: the nu sums in registers per owned row part, the xi sums through LDS (the staging buffers, two 32-channel halves): eight partial
//     row tiles per block, every thread then combines them for whole 16-byte channel groups and stores NHWC channel runs with scale / bias / residual / ReLU.
// The per-output arithmetic depends only on the tile's own pixels: results do not depend on the batch position.
#include "common.h"
#include "winograd_pack.h"

#include <atomic>
#include <type_traits>

namespace vatl {

struct F4Params {
    const float* x;
    const float* u;
    const float* scale;
    const float* bias;
    const float* res;
    float* y;
    // training epilogues (MODE 1: BatchNorm batch statistics of the raw output; MODE 2: BatchNorm-backward fusion of a data-gradient launch, semantics of
    // ConvParams::bz .. in conv_igemm.hip): (sum, sum^2) / (sum g, sum g * xhat) per (16-tile row block, channel) as doubles, the layout vatl_bn_train_finalize /
    // vatl_bn_bwd_from_stats reduce in a fixed order
    double* stats;
    const float* bz;
    const float* bmy;
    const float* bsc;
    const float* bbi;
    const float* bmu;
    const float* bis;
    int N, H, W, Cin, Cout;
    int TH, TW, tpi, Mtiles, m_tiles, n_tiles, stages, relu;
    int xper;                          // XCDs per 64-channel slice (8 / n_tiles), 0: plain order
    int chunk;                         // tile groups per XCD of a slice: ceil(m_tiles / xper)
    unsigned x_bytes, y_bytes, u_bytes;
    FastDivU d_tpi, d_TW;
};

typedef __attribute__((address_space(3))) void f4_lds_void;
constexpr int F4_TB = 16;                                  // tiles per block
constexpr int F4_NDMA = 36;                                // wave-wide DMA instructions per stage (1 KB each): one per patch pixel, lane = (tile, channel quad)
constexpr int F4_STAGE = F4_NDMA * 256;                    // floats per stage buffer: 9216 (36 864 bytes)
constexpr int F4_KDMA = F4_NDMA / 4;                       // per wave: 9
constexpr unsigned F4_OOB = 0xFFFF0000u;                   // stays out of range with stage * 64 bytes added (tensors <= 0xFFFF0000 bytes: checked on the host)
constexpr int F4_ZT = 148;                                 // floats per (slot, tile) of the output exchange: 4 rows of 36 (32 channels + 4) + 4: four tiles apart = 16 banks apart
constexpr int F4_LDS_FLOATS = 8 * 16 * F4_ZT;              // the output exchange (18 944 floats) is a little larger than the two stage buffers (18 432)

// B^T (and, for the columns, B) along one axis: combination XI of six values
template <int XI, typename T>
__device__ __forceinline__ T f4_comb(const T& d0, const T& d1, const T& d2, const T& d3, const T& d4, const T& d5) {
    if constexpr (XI == 0) return 4.f * d0 - 5.f * d2 + d4;
    else if constexpr (XI == 1) return (d3 + d4) - 4.f * (d1 + d2);
    else if constexpr (XI == 2) return (d4 - d3) + 4.f * (d1 - d2);
    else if constexpr (XI == 3) return (d4 - d2) + 2.f * (d3 - d1);
    else if constexpr (XI == 4) return (d4 - d2) - 2.f * (d3 - d1);
    else return 4.f * d1 - 5.f * d3 + d5;
}

// rows of the 6 x 6 patch that combination XI reads (the others have a zero coefficient)
template <int XI> __device__ __forceinline__ constexpr bool f4_uses(int i) {
    return XI == 0 ? (i == 0 || i == 2 || i == 4) : (XI == 5 ? (i == 1 || i == 3 || i == 5) : (i >= 1 && i <= 4));
}

__device__ __forceinline__ f32x4 f4_buf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}
// F4_ABL (tools/f4_ablate.sh only; wrong results): 1 no filter loads, 2 no LDS reads / row combinations, 4 no staging DMA, 8 no stage barrier,
// 16 no output transform / write-out (the accumulators stay alive), 32 no wait for the block's first stage and first filter fragments
#ifndef F4_ABL
#define F4_ABL 0
#endif
#ifndef F4_STAGGER
#define F4_STAGGER 0
#endif
// Filter fragments and the staging DMA share the wave's IN-ORDER vector-memory counter, and hipcc does not count an LDS-DMA request (the builtin below) in the
// bookkeeping behind its own waits: a fragment load issued after a DMA request completes after it, and the compiler's wait for an OLDER batch ("the four younger loads
// may stay in flight": vmcnt(4)) in fact also waits for a request issued in between.  With one request behind every position's MFMAs every position waited for a
// fresh L2 / HBM round trip (tools/f4_ablate.sh: the requests cost 21 % of the kernel, the fragment loads 16 %).  The requests of a stage therefore go out in ONE burst
// at the point of the stage from which the next wait on a younger fragment batch is farthest away (f4_part, BURST).
__device__ __forceinline__ f32x4 f4_filter_load(__amdgpu_buffer_rsrc_t r, unsigned off) {
    if constexpr ((F4_ABL & 1) != 0) { f32x4 v; asm volatile("" : "=v"(v)); return v; }
    else return f4_buf_load4(r, off);
}
// compile-time loop index
template <int I, int N, typename F>
__device__ __forceinline__ void f4_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        f4_static_for<I + 1, N>(f);
    }
}
// One row part of a wave: row XI of the transform domain, columns NU0 .. NU0 + NUN - 1, accumulators PL0 .. PL0 + NUN - 1.
// Pl: this lane's slot in the current stage buffer (+ 256 floats per patch pixel); ub: byte offset of (stage, position 0, this block's first column block, this lane).
// BURST: the position of the STAGE (0 .. 8, -1: none in this part) behind whose first four MFMAs the next stage's DMA requests go out.
// B: ring of three fragment sets: position pl of a stage multiplies with B[pl % 3]; the set of position pl + 2 is requested at position pl (nine positions per stage:
// the ring index carries over from stage to stage).  POS(k): transform-domain position (xi * 6 + nu) of the wave's k-th position of a stage.
template <int XI, int NU0, int NUN, int PL0, int BURST, int NB, bool FIRST, typename DMA, typename POS>
__device__ __forceinline__ void f4_part(const float* Pl, f32x4 (&acc)[9][NB], __amdgpu_buffer_rsrc_t ur, unsigned ub, unsigned pos_bytes, unsigned stage_bytes, f32x4 (&B)[3][NB],
                                        DMA dma_burst, bool more, POS pos_of) {
    // row combination XI of the six patch columns as a chain of multiply-adds over the rows it uses, term by term (18 - 24 terms: column-major), the reads a ring of
    // DEPTH terms ahead: ~DEPTH x 16 cycles of vector work cover an LDS read (with the read one term ahead every term waited ~50 cycles for its operand: 30 % of the
    // wave time at s_waitcnt), and four 16-byte temporaries are what the registers next to 144 accumulators allow (the compiler, left alone, issues all reads of the
    // part first: ~100 registers).  One instruction more per column than the shared-subexpression form for rows 1 - 4.
    constexpr float CF[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0}, {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
    constexpr int NR = (XI == 0 || XI == 5) ? 3 : 4, R0 = XI == 0 ? 0 : 1, RS = (XI == 0 || XI == 5) ? 2 : 1;      // rows R0, R0 + RS, ...
    constexpr int NT = 6 * NR, DEPTH = 4;
    f32x4 R[6];
    f32x4 ring[DEPTH];
    auto term_ptr = [&](int t) { return reinterpret_cast<const f32x4*>(Pl + ((F4_ABL & 2) ? 0 : ((R0 + (t % NR) * RS) * 6 + t / NR) * 256)); };
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) ring[t] = *term_ptr(t);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int jj = t / NR, k = t % NR;
        const f32x4 cur = ring[t % DEPTH];
        if (t + DEPTH < NT) ring[t % DEPTH] = *term_ptr(t + DEPTH);
        const float c = CF[XI][R0 + k * RS];
        if (k == 0) R[jj] = c == 1.f ? cur : c * cur;
        else if (c == 1.f) R[jj] += cur;
        else if (c == -1.f) R[jj] -= cur;
        else R[jj] += c * cur;
        __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    f4_static_for<0, NUN>([&](auto qc) {
        constexpr int q = decltype(qc)::value, nu = NU0 + q;
        const f32x4 V = f4_comb<nu>(R[0], R[1], R[2], R[3], R[4], R[5]);        // column combination nu of the row values
        constexpr int pl = PL0 + q;
        f32x4 (&bcur)[NB] = B[pl % 3];
        auto prefetch = [&] {                              // the filter fragments of the position after next (of this stage, or the first / second one of the next stage)
            constexpr int t = pl + 2;
            if (t < 9 || more) {
                const unsigned nb = t < 9 ? ub + (unsigned)pos_of(t) * pos_bytes : ub + stage_bytes + (unsigned)pos_of(t - 9) * pos_bytes;
#pragma unroll
                for (int n = 0; n < NB; ++n) B[t % 3][n] = f4_filter_load(ur, nb + n * 1024u);
            }
        };
        if constexpr (pl != BURST) prefetch();             // ... requested before this position's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[pl][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[n][0], V[0], FIRST ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[pl][n], 0, 0, 0);   // FIRST: the C operand is the inline constant 0 (no 144 v_mov to clear the accumulators)
        if constexpr (pl == BURST) {
            // the burst position: its own fragments and the next position's are older than the requests, so nothing here waits for them; the first younger batch
            // (position pl + 2) is waited for 12 MFMAs + the next part's row combinations + one whole position later
            __builtin_amdgcn_sched_barrier(0);
            dma_burst();
            prefetch();
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 1; s < 4; ++s)
#pragma unroll
            for (int n = 0; n < NB; ++n) acc[pl][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bcur[n][s], V[s], acc[pl][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    });
}

// nu sums of one row part: Z[b] = sum over the part's columns of A^T[b][nu] * M[nu], for the column block pair H (two 16-channel blocks) -> LDS slot SLOT.
// The MFMAs run with the FILTER fragment as their A operand (rows = output channels) and the transformed input as B (columns = tiles), so a lane's accumulator
// quad is four CONSECUTIVE CHANNELS (16 n + 4 (lane / 16) + e) of one tile (lane % 16): each z[b] goes to the exchange area as ONE 16-byte write — the layout the
// readers and the NHWC stores want.  (Round 5 had the operands the other way round — a quad = four tiles of one channel — and scattered every value with a 4-byte
// write: 16 ds_write_b32 + the register shuffles around them per z[b] quad.  Same products, same order of the four k-steps: bit-identical.)
template <int NU0, int NUN, int PL0, int SLOT, int NB>
__device__ __forceinline__ void f4_nu_sums(const f32x4 (&acc)[9][NB], float* Zs, int h, int lane) {
    // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1] through the sums and differences of the column pairs (1, 2) and (3, 4) — the form the readers
    // use along the other axis: 10 vector operations for a full row instead of 18, 3 / 5 for the two kinds of half row (round 6; round 5 summed term by term from
    // zero, so the two agree to fp32 rounding, not bit for bit)
    static_assert((NU0 == 0 && (NUN == 3 || NUN == 6)) || (NU0 == 3 && NUN == 3), "row parts: columns 0-2, 3-5 or 0-5");
    const int tl = lane & 15, cg = lane >> 4;
    float* dst = Zs + (SLOT * 16 + tl) * F4_ZT + cg * 4;
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
        const int n = 2 * h + nn;
        f32x4 z0, z1, z2, z3;
        if constexpr (NU0 == 0) {
            const f32x4 s12 = acc[PL0 + 1][n] + acc[PL0 + 2][n], d12 = acc[PL0 + 1][n] - acc[PL0 + 2][n];
            z0 = acc[PL0][n] + s12; z1 = d12; z2 = s12; z3 = d12;
            if constexpr (NUN == 6) {
                const f32x4 s34 = acc[PL0 + 3][n] + acc[PL0 + 4][n], d34 = acc[PL0 + 3][n] - acc[PL0 + 4][n];
                z0 += s34; z1 += 2.f * d34; z2 += 4.f * s34; z3 += 8.f * d34 + acc[PL0 + 5][n];
            }
        } else {
            const f32x4 s34 = acc[PL0][n] + acc[PL0 + 1][n], d34 = acc[PL0][n] - acc[PL0 + 1][n];
            z0 = s34; z1 = 2.f * d34; z2 = 4.f * s34; z3 = 8.f * d34 + acc[PL0 + 2][n];
        }
        *reinterpret_cast<f32x4*>(dst + 0 * 36 + nn * 16) = z0;
        *reinterpret_cast<f32x4*>(dst + 1 * 36 + nn * 16) = z1;
        *reinterpret_cast<f32x4*>(dst + 2 * 36 + nn * 16) = z2;
        *reinterpret_cast<f32x4*>(dst + 3 * 36 + nn * 16) = z3;
    }
}

// MODE 0: y = act(scale * conv + bias + residual) (inference); 1: y = conv, batch statistics; 2: y = (conv + residual) * [consumer's ReLU mask], BatchNorm-backward sums
// NB: 16-channel column blocks per block (4 = 64 output channels; 2 was measured for HRNet's 32-channel branch and is not instantiated: see _supported)
template <int WV, int MODE, int NB>
__device__ __forceinline__ void f4_body(const F4Params& p, float* smem, int m_tile, int n_tile) {
    const int tid = threadIdx.x, lane = tid & 63;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, p.u_bytes, 0x00020000);

    // ---- staging geometry: DMA instruction px (= patch pixel (px / 6, px % 6)) of a stage writes the 1 KB row px of the stage buffer; its lane L delivers slot L of
    // that row = (tile L / 4, channel quad (L % 4 - 2 (tile / 8)) & 3): with that rotation the 16 lanes of every ds_read_b128 service group ({0-3, 12-15, 20-27}, ...:
    // tiles 0-3 / 12-15 of one quad and tiles 4-11 of the next) hit 16 different 16-byte bank slots (a rotation by tile / 4 left every read a 2-way conflict:
    // SQ_LDS_BANK_CONFLICT 48 % of the LDS cycles).  Per lane that is ONE
    // base offset (patch pixel (0, 0) of its tile, its quad) + a wave-uniform pixel offset, and a 6 + 6 bit row / column mask of the pixels inside the image ----
    unsigned tile_base, inside;
    {
        const int tl = lane >> 2, qd = ((lane & 3) - 2 * (tl >> 3)) & 3;
        const int T = m_tile * F4_TB + tl;
        const int img = fdiv(T, p.d_tpi), rem = T - img * p.tpi;
        const int ty = fdiv(rem, p.d_TW), tx = rem - ty * p.TW;
        tile_base = (unsigned)((((img * p.H + 4 * ty - 1) * p.W + 4 * tx - 1) * p.Cin + qd * 4) * 4);     // (may wrap below zero for the first tile: its row / column -1 is masked)
        inside = 0;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int iy = 4 * ty - 1 + i, ix = 4 * tx - 1 + i;
            inside |= (T < p.Mtiles && iy >= 0 && iy < p.H ? 1u : 0u) << i;
            inside |= (ix >= 0 && ix < p.W ? 1u : 0u) << (6 + i);
        }
    }
    const unsigned row_bytes = (unsigned)(p.W * p.Cin * 4), px_bytes = (unsigned)(p.Cin * 4);
    auto request_piece = [&](int stage, float* buf, int k) {                            // k: compile-time at every call site
        unsigned tb = tile_base, in = inside;
        asm volatile("" : "+v"(tb), "+v"(in));             // (the offset is formed HERE: hoisted out of the stage loop the nine of them cost nine registers next to 144 accumulators)
        const int px = WV + 4 * k, i = px / 6, jx = px - 6 * i;
        const bool ok = ((in >> i) & (in >> (6 + jx)) & 1u) != 0;
        const unsigned off = ok ? tb + (unsigned)i * row_bytes + (unsigned)jx * px_bytes : F4_OOB;
        if constexpr ((F4_ABL & 4) == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (f4_lds_void*)(buf + px * 256), 16, off, (unsigned)stage * 64u, 0, 0);
    };

    // ---- operands ----
    const int t16 = lane & 15, quad = lane >> 4;
    const int lane_floats = (t16 * 4 + ((quad + 2 * (t16 >> 3)) & 3)) * 4;     // this lane's slot in every pixel row of a stage buffer
    const unsigned nbg = (unsigned)(p.Cout >> 4);
    const unsigned pos_bytes = nbg * 1024u;                // bytes between consecutive positions of U
    const unsigned stage_bytes = 36u * pos_bytes;
    const unsigned ublock = ((unsigned)n_tile * (unsigned)NB * 64u + (unsigned)lane) * 16u;
    // the wave's two row parts, the half row first (its reads are done a third into the stage: the next stage's requests go out behind the SECOND part's reads)
    constexpr int XA = WV == 0 ? 1 : (WV == 1 ? 1 : 4), NA0 = (WV == 0 || WV == 2) ? 0 : 3;          // half row: 3 positions
    constexpr int XB = WV == 0 ? 0 : (WV == 1 ? 2 : (WV == 2 ? 3 : 5));                               // full row: 6 positions
    f32x4 acc[9][NB];

#pragma unroll
    for (int k = 0; k < F4_KDMA; ++k) request_piece(0, smem, k);
    auto pos_of = [](int k) { return k < 3 ? XA * 6 + NA0 + k : XB * 6 + (k - 3); };     // the wave's k-th position of a stage
    f32x4 B[3][NB];
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int n = 0; n < NB; ++n) B[k][n] = f4_filter_load(ur, ublock + (unsigned)pos_of(k) * pos_bytes + n * 1024u);
    auto stage = [&](int s, auto first) {
        constexpr bool FIRST = decltype(first)::value;
        // This wave's pieces of stage s have landed.  The vector-memory counter retires IN ORDER, and the 2 NB youngest operations at this point are the filter
        // fragments of the stage's first two positions (requested behind the last two positions of the stage before, or in the prologue) — the staging requests are
        // all older.  Waiting for everything BUT those 2 NB leaves the fragments in flight across the barrier; the MFMAs that use them wait for them on their own
        // (hipcc tracks buffer loads into registers).  Round 5 waited with vmcnt(0): every stage began with a full L2 round trip for fragments requested ~0.5 us earlier.
        if ((F4_ABL & 32) == 0 || s > 0) {
            if constexpr ((F4_ABL & 1) != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else { static_assert(NB == 4, "the wait count below is 2 * NB"); asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
        }
        if constexpr ((F4_ABL & 8) == 0) __builtin_amdgcn_s_barrier();                      // ... everyone's; and every wave is done reading the other buffer (stage s - 1)
        asm volatile("" ::: "memory");
        float* cur = smem + (s & 1) * F4_STAGE;
        float* nxt = smem + ((s + 1) & 1) * F4_STAGE;
        const float* Pl = cur + lane_floats;
        const unsigned ub = ublock + (unsigned)s * stage_bytes;
        const bool more = s + 1 < p.stages;
        auto burst = [&] {                                 // (the other buffer: every wave is past this stage's barrier, i.e. done with it)
            if (more) {
#pragma unroll
                for (int k = 0; k < F4_KDMA; ++k) request_piece(s + 1, nxt, k);
            }
        };
        f4_part<XA, NA0, 3, 0, 2, NB, FIRST>(Pl, acc, ur, ub, pos_bytes, stage_bytes, B, burst, more, pos_of);
        f4_part<XB, 0, 6, 3, -1, NB, FIRST>(Pl, acc, ur, ub, pos_bytes, stage_bytes, B, burst, more, pos_of);
    };
    stage(0, std::true_type{});                            // (Cin >= 64: at least four stages)
    for (int s = 1; s < p.stages; ++s) stage(s, std::false_type{});

    // ---- output transform ----
    if constexpr ((F4_ABL & 16) != 0) {
        f32x4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 9; ++q)
#pragma unroll
            for (int n = 0; n < NB; ++n) t += acc[q][n];
        if (t[0] + t[1] + t[2] + t[3] == 12345.678f) p.y[0] = t[0];
        return;
    }
    lds_barrier();                                         // every wave is done with the stage buffers
    float* Zs = smem;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0, p.res ? p.y_bytes : 0u, 0x00020000);
    const float lo = p.relu ? 0.f : -INFINITY;
    const int c4 = tid & 7;                                // this thread's 16-byte channel group of the 32-channel half (the same for both of its items)
#pragma unroll
    for (int h = 0; h < NB / 2; ++h) {
        f4_nu_sums<NA0, 3, 0, 2 * WV, NB>(acc, Zs, h, lane);
        f4_nu_sums<0, 6, 3, 2 * WV + 1, NB>(acc, Zs, h, lane);
        lds_barrier();
        const int n = n_tile * (16 * NB) + h * 32 + c4 * 4;
        const f32x4 one = {1.f, 1.f, 1.f, 1.f}, nul = {0.f, 0.f, 0.f, 0.f};
        const f32x4 sc = (MODE == 0 && p.scale) ? *reinterpret_cast<const f32x4*>(p.scale + n) : one;
        const f32x4 bi = (MODE == 0 && p.bias) ? *reinterpret_cast<const f32x4*>(p.bias + n) : nul;
        f32x4 mu = nul, is = nul, msc = nul, mbi = one;    // MODE 2 (no mask given: 0 * z + 1 > 0)
        __amdgpu_buffer_rsrc_t zr = yr, mr = yr;
        if constexpr (MODE == 2) {
            mu = *reinterpret_cast<const f32x4*>(p.bmu + n); is = *reinterpret_cast<const f32x4*>(p.bis + n);
            if (p.bsc) { msc = *reinterpret_cast<const f32x4*>(p.bsc + n); mbi = *reinterpret_cast<const f32x4*>(p.bbi + n); }
            zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bz), 0, p.y_bytes, 0x00020000);
            mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bmy ? p.bmy : p.bz), 0, p.bmy ? p.y_bytes : 0u, 0x00020000);
        }
        f32x4 s1 = nul, s2 = nul;                          // this thread's share of the two per-channel sums (MODE 1 / 2)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int id = tid + 256 * r;                  // (tile, b, c4)
            const int b = (id >> 3) & 3, tl = id >> 5;
            const int T = m_tile * F4_TB + tl;
            const int img = fdiv(T, p.d_tpi), rem = T - img * p.tpi;
            const int ty = fdiv(rem, p.d_TW), tx = rem - ty * p.TW;
            const bool ok = T < p.Mtiles;
            const unsigned base = ok ? (unsigned)((((img * p.H + 4 * ty) * p.W + 4 * tx + b) * p.Cout + n) * 4) : 0xFFFFFFF0u;
            const unsigned rowb = (unsigned)(p.W * p.Cout * 4);
            f32x4 rs[4], zt[4], yt[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const unsigned off = ok ? base + a * rowb : 0xFFFFFFF0u;
                rs[a] = MODE == 1 ? nul : f4_buf_load4(rr, off);
                if constexpr (MODE == 2) { zt[a] = f4_buf_load4(zr, off); yt[a] = f4_buf_load4(mr, off); }
            }
            f32x4 S[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) S[k] = *reinterpret_cast<const f32x4*>(Zs + (k * 16 + tl) * F4_ZT + b * 36 + c4 * 4);
            // slots: 0 = row 1 (nu 0-2), 1 = row 0, 2 = row 1 (nu 3-5), 3 = row 2, 4 = row 4 (nu 0-2), 5 = row 3, 6 = row 4 (nu 3-5), 7 = row 5
            const f32x4 z0 = S[1], z1 = S[0] + S[2], z2 = S[3], z3 = S[5], z4 = S[4] + S[6], z5 = S[7];
            const f32x4 d12 = z1 - z2, s12 = z1 + z2, d34 = z3 - z4, s34 = z3 + z4;
            f32x4 o[4];
            o[0] = z0 + s12 + s34;
            o[1] = d12 + 2.f * d34;
            o[2] = s12 + 4.f * s34;
            o[3] = d12 + 8.f * d34 + z5;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                f32x4 v;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if constexpr (MODE == 0) v[c] = fmaxf(fmaf(o[a][c], sc[c], bi[c]) + rs[a][c], lo);
                    else if constexpr (MODE == 1) { v[c] = o[a][c]; s1[c] += v[c]; s2[c] += v[c] * v[c]; }       // tiles past the end: exact zeros
                    else {
                        const float d = o[a][c] + rs[a][c];
                        const bool on = p.bmy ? yt[a][c] > 0.f : fmaf(zt[a][c], msc[c], mbi[c]) > 0.f;
                        v[c] = on ? d : 0.f;
                        s1[c] += v[c];
                        s2[c] += v[c] * ((zt[a][c] - mu[c]) * is[c]);
                    }
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(__attribute__((ext_vector_type(4))) unsigned, v), yr, ok ? base + a * rowb : 0xFFFFFFF0u, 0, 0);
            }
        }
        if constexpr (MODE != 0) {
            // per-channel sums of this block's 16 tiles x 16 pixels: the 8 lanes of a wave that share c4 (lane bits 3 .. 5) by shuffles in fp32 (64 values each), the four
            // waves through LDS and in double, one (row block = m_tile, channel) pair per channel of the half: fixed order, no atomics
#pragma unroll
            for (int sh = 8; sh < 64; sh <<= 1)
#pragma unroll
                for (int c = 0; c < 4; ++c) { s1[c] += __shfl_xor(s1[c], sh, 64); s2[c] += __shfl_xor(s2[c], sh, 64); }
            lds_barrier();                                 // every thread is done reading the exchange area
            if (lane < 8) {
                *reinterpret_cast<f32x4*>(Zs + (WV * 8 + lane) * 8) = s1;
                *reinterpret_cast<f32x4*>(Zs + (WV * 8 + lane) * 8 + 4) = s2;
            }
            lds_barrier();
            if (tid < 32) {
                double a1 = 0.0, a2 = 0.0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { a1 += (double)Zs[(w * 8 + (tid >> 2)) * 8 + (tid & 3)]; a2 += (double)Zs[(w * 8 + (tid >> 2)) * 8 + 4 + (tid & 3)]; }
                const long long ch = (long long)m_tile * p.Cout + n_tile * (16 * NB) + h * 32 + tid;
                p.stats[ch * 2 + 0] = a1;
                p.stats[ch * 2 + 1] = a2;
            }
        }
        if (h + 1 < NB / 2) lds_barrier();                 // the second half overwrites the exchange area
    }
}

template <int MODE, int NB>
__global__ __launch_bounds__(256, 2) void winograd_f4_kernel(F4Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int bid = blockIdx.x;
    int m_tile, n_tile;
    if (p.xper > 0) {                                      // block b runs on XCD b % 8: every XCD keeps ONE 64-channel slice of U in its L2
        const int xcd = bid & 7, s = bid >> 3;
        n_tile = xcd % p.n_tiles;
        // ... and a CONTIGUOUS run of tile groups: the blocks resident on an XCD at one time are then neighbours in the image (64 consecutive groups = 5 images at
        // 64 x 48), so the two halo rows / columns a 6 x 6 patch shares with the tiles around it, and the second half of every 128-byte line that the next
        // 16-channel stage reads, are found in THAT XCD's L2.  (Round 5 dealt the groups round-robin over the XCDs, m_tile = s * xper + xcd / n_tiles: vertical
        // neighbours sat on different XCDs and the conv launches fetched 3.7x their input, profiles/r05_pmc_summary.json.)
        m_tile = (xcd / p.n_tiles) * p.chunk + s;
        if (s >= p.chunk) return;
    } else {
        n_tile = bid % p.n_tiles;
        m_tile = bid / p.n_tiles;
    }
    if (m_tile >= p.m_tiles) return;
#if F4_STAGGER
    // experiment (tools/f4_ablate.sh stagger): the blocks of a launch start together, take the same time and are replaced together, so the whole chip loads, computes and
    // stores in phase; the SECOND block of every CU (ids 256 .. 511 of the first round) starts F4_STAGGER x 8128 cycles late, and the offset survives the rounds
    if (bid >= 256 && bid < 512) {
#pragma unroll 1
        for (int i = 0; i < F4_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    switch (wave) {                                        // the position split is per wave and compile-time: four instantiations of the body
        case 0: f4_body<0, MODE, NB>(p, smem, m_tile, n_tile); break;
        case 1: f4_body<1, MODE, NB>(p, smem, m_tile, n_tile); break;
        case 2: f4_body<2, MODE, NB>(p, smem, m_tile, n_tile); break;
        default: f4_body<3, MODE, NB>(p, smem, m_tile, n_tile); break;
    }
}

// (Cout, Cin, 3, 3) -> U = G g G^T (float64 arithmetic) in fragment order (winograd_pack.h: f4_pack_item)
__global__ __launch_bounds__(256) void winograd_f4_pack_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin, int dgrad) {
    f4_pack_item(w, u, dgrad, dgrad ? Cout : Cin, Cout, Cin, blockIdx.x * 256LL + threadIdx.x);
}

}  // namespace vatl

using namespace vatl;

extern "C" int64_t vatl_winograd_f4_weight_floats(int Cout, int Cin) { return 36LL * Cout * Cin; }

extern "C" int vatl_conv3x3_winograd_f4_supported(int N, int H, int W, int Cin, int Cout) {
    // (32 -> 32 channels with 32-channel blocks, NB = 2, was built and measured: 384 us against 405 us for winograd_c32 on HRNet's first branch at 1024 crops — two stages
    //  per block leave the transforms and the write-out unamortised; not served)
    if (N <= 0 || H < 4 || W < 4 || (H & 3) || (W & 3) || Cin < 64 || (Cin & 15) || Cout < 64 || (Cout & 63)) return 0;
    const long long xe = (long long)N * H * W * Cin, ye = (long long)N * H * W * Cout;
    return xe * 4 <= (long long)F4_OOB && ye < (1LL << 30) && 36LL * Cin * Cout < (1LL << 28) ? 1 : 0;
}

extern "C" int64_t vatl_winograd_f4_stats_row_blocks(int64_t N, int H, int W) { return (N * (H / 4) * (W / 4) + F4_TB - 1) / F4_TB; }

// (Cout, Cin) of the PACKED filter: with data_gradient the source is the forward filter (O = Cin, I = Cout, 3, 3) and the result the filter of dX = conv(dY, rot180(w)^T)
extern "C" int vatl_pack_winograd_f4_weight(const float* w, float* u, int Cout, int Cin, int data_gradient, void* stream) {
    if (!w || !u || Cout <= 0 || Cin <= 0 || (Cout & 15) || (Cin & 15)) return fail(VATL_EINVAL, "pack_winograd_f4_weight: needs channel counts that are multiples of 16");
    const long long total = (long long)(Cin >> 4) * 36 * (Cout >> 4) * 64;
    hipLaunchKernelGGL(winograd_f4_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, u, Cout, Cin, data_gradient ? 1 : 0);
    return check_launch("winograd_f4_pack");
}

struct F4Fuse {
    const float *z, *mask_y, *scale, *bias, *mean, *invstd;
};

static int f4_impl(int mode, const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y, double* stats, int64_t* row_blocks_used,
                   const F4Fuse* fuse, int N, int H, int W, int Cin, int Cout, int relu, void* stream) {
    if (!x || !u || !y) return fail(VATL_EINVAL, "conv3x3_winograd_f4: null pointer");
    if (!vatl_conv3x3_winograd_f4_supported(N, H, W, Cin, Cout))
        return fail(VATL_EINVAL, "conv3x3_winograd_f4: serves H, W multiples of 4, Cin >= 64 a multiple of 16, Cout a multiple of 64 (got %d x %d, %d -> %d)", H, W, Cin, Cout);
    if (mode != 0 && !stats) return fail(VATL_EINVAL, "conv3x3_winograd_f4: the training epilogues need a statistics buffer");
    if (mode == 2 && (!fuse || !fuse->z || !fuse->mean || !fuse->invstd || (!fuse->scale != !fuse->bias)))
        return fail(VATL_EINVAL, "conv3x3_winograd_f4_fwd_bnbwd: needs the consumer layer's conv output and saved statistics (and scale WITH bias, or neither)");
    F4Params p{};
    p.x = x; p.u = u; p.scale = scale; p.bias = bias; p.res = residual; p.y = y; p.stats = stats;
    if (fuse) { p.bz = fuse->z; p.bmy = fuse->mask_y; p.bsc = fuse->scale; p.bbi = fuse->bias; p.bmu = fuse->mean; p.bis = fuse->invstd; }
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.relu = relu;
    p.TH = H / 4; p.TW = W / 4; p.tpi = p.TH * p.TW;
    const long long mt = (long long)N * p.tpi;
    if (mt >= (1LL << 30)) return fail(VATL_EINVAL, "conv3x3_winograd_f4: too many tiles");
    p.Mtiles = (int)mt;
    p.m_tiles = cdiv(mt, F4_TB);
    p.n_tiles = Cout / 64;
    p.stages = Cin / 16;
    p.x_bytes = (unsigned)((long long)N * H * W * Cin * 4); p.y_bytes = (unsigned)((long long)N * H * W * Cout * 4);
    p.u_bytes = (unsigned)(36LL * Cin * Cout * 4);
    p.d_tpi = make_fastdiv((unsigned)p.tpi); p.d_TW = make_fastdiv((unsigned)p.TW);
    if (row_blocks_used) *row_blocks_used = p.m_tiles;
    long long grid;
    if (p.n_tiles == 1 || p.n_tiles == 2 || p.n_tiles == 4 || p.n_tiles == 8) {
        p.xper = 8 / p.n_tiles;
        p.chunk = cdiv(p.m_tiles, p.xper);
        grid = 8LL * p.chunk;
    } else {
        p.xper = 0;
        grid = (long long)p.m_tiles * p.n_tiles;
    }
    if (grid >= (1LL << 31)) return fail(VATL_EINVAL, "conv3x3_winograd_f4: too many blocks");
    const int smem = F4_LDS_FLOATS * (int)sizeof(float);
    static std::atomic<unsigned> configured[3] = {{0}, {0}, {0}};
    const void* kern = mode == 0 ? (const void*)winograd_f4_kernel<0, 4> : (mode == 1 ? (const void*)winograd_f4_kernel<1, 4> : (const void*)winograd_f4_kernel<2, 4>);
    if (int rc = ensure_dynamic_lds(kern, smem, configured[mode], "winograd_f4")) return rc;
    if (mode == 0) hipLaunchKernelGGL((winograd_f4_kernel<0, 4>), dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream, p);
    else if (mode == 1) hipLaunchKernelGGL((winograd_f4_kernel<1, 4>), dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((winograd_f4_kernel<2, 4>), dim3((unsigned)grid), dim3(256), smem, (hipStream_t)stream, p);
    meter_add(1, 2.0 * ((double)p.m_tiles * F4_TB) * (double)Cout * (double)Cin * 36.0);
    meter_route(mode == 2 ? kRouteWinoF4BnBwd : kRouteWinoF4);
    return check_launch("winograd_f4");
}

extern "C" int vatl_conv3x3_winograd_f4_fwd(const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y, int N, int H, int W,
                                            int Cin, int Cout, int relu, void* stream) {
    return f4_impl(0, x, u, scale, bias, residual, y, nullptr, nullptr, nullptr, N, H, W, Cin, Cout, relu, stream);
}

extern "C" int vatl_conv3x3_winograd_f4_fwd_stats(const float* x, const float* u, float* y, double* stats, int64_t* row_blocks_used, int N, int H, int W, int Cin, int Cout,
                                                  void* stream) {
    return f4_impl(1, x, u, nullptr, nullptr, nullptr, y, stats, row_blocks_used, nullptr, N, H, W, Cin, Cout, 0, stream);
}

extern "C" int vatl_conv3x3_winograd_f4_fwd_bnbwd(const float* x, const float* u, const float* residual, float* y, int N, int H, int W, int Cin, int Cout, const float* bn_z,
                                                  const float* bn_mask_y, const float* bn_scale, const float* bn_bias, const float* bn_mean, const float* bn_invstd,
                                                  double* stats, int64_t* row_blocks_used, void* stream) {
    const F4Fuse f{bn_z, bn_mask_y, bn_scale, bn_bias, bn_mean, bn_invstd};
    return f4_impl(2, x, u, nullptr, nullptr, residual, y, stats, row_blocks_used, &f, N, H, W, Cin, Cout, 0, stream);
}
