// Implicit-GEMM convolution forward on the gfx950 fp32 matrix cores.
//
//   Y[m][n] = act( (sum_k A[m][k] * Wt[n][k]) * scale[n] + bias[n] (+ R[m][n]) )
//   m = (image b, out row oy, out col ox)   M = N*Ho*Wo      (GEMM rows = pixels)
//   n = output channel                                        (GEMM cols)
//   k = (r, s, c) filter tap x input channel, c fastest       K = R*S*Cin
//
// Activations are NHWC, so for a fixed tap (r,s) a 32-wide k-tile is 128
// contiguous bytes of one input pixel: the A operand is gathered on the fly
// (zero-filled outside the image), never materialised (no im2col buffer).
// Weights are pre-packed [CoutPad][K] (vatl_pack_conv_weight) so both operands
// are K-contiguous and staged with 16-byte loads.
//
// Arithmetic: v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulate) —
// the reference path is fp32 and parity is 1e-4 relative through 50+ layers,
// which bf16 inputs do not hold (SURVEY.md §7 hard part 1).  Peak for this
// instruction is 157.3 TFLOP/s (MI355X_MICROARCH.md).
//
// Block = 256 threads = 4 waves, tile BM x BN x 32, wave tile WM x WN made of
// 32x32 MFMA tiles.  K-loop: LDS double buffer, one barrier per k-tile; the next
// tile's global loads are issued before the current tile's 16 k-steps of MFMAs
// and written to the other LDS buffer after them (register staging), so HBM/L2
// latency hides under >= 2048 cycles of matrix work per wave.
// LDS rows are exactly one k-tile (32 floats = 128 bytes), bank conflicts are avoided by an XOR swizzle of
// the 16-byte chunk position: chunk c of tile row r lives at position c ^ ((r >> 1) & 7).  A ds_read_b128
// fragment read (16 consecutive rows at one logical chunk per lane group) then touches all 16 chunk slots of
// the 256-byte bank row, and a ds_write_b128 staging pass (8 rows x 8 chunks per wave) still covers whole
// rows: both conflict-free.  Against the 36-float padded rows of the first version this saves 11 % of LDS:
// the 64x128 and 128x64 tiles drop from 55 KB to 48 KB per block = THREE resident blocks per CU instead of two
// (their 132-138 VGPRs allow three waves per SIMD), which is what small-batch launches (fine-tune steps: one to
// three rounds of resident blocks) need to fill the chip.
//
// One ds_read_b128 feeds four MFMA k-steps: lanes 0-31 hold k = 8g+t, lanes
// 32-63 hold k = 8g+4+t at step t (same permutation for A and B, so the sum
// over the 8 k's of group g is complete after t = 0..3).  The per-output
// reduction order is fixed and independent of the batch position (needed for
// the THC de-duplication, SURVEY.md §7 hard part 3).
//
// The same kernel runs ConvTranspose2d(4,2,1) as four 2x2 sub-pixel phases
// (blockIdx.y = py*2+px): pad = (1-py, 1-px), output scattered to (2y+py, 2x+px).
#include "common.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>

namespace vatl {

struct ConvParams {
    const float* x;
    const float* w;
    const float* scale;
    const float* bias;
    const float* res;
    float* y;
    int N, H, W, Cin;
    int Cout, CoutPad;
    int R, S, stride, pad_y, pad_x;
    int Ho, Wo, M;
    int OH, OW, osy, osx, ooy, oox;   // output pixel = (oy*osy+ooy, ox*osx+oox) in an OH x OW image
    int relu, out_nchw, deconv;
    int kpr;                          // k-tiles per filter tap  (Cin/32; 1 for the stem)
    int ktiles;                       // total k-tiles
    int n_tiles, m_tiles;
    int order;                        // tile order inside an XCD's run: 0 n-tile fastest, 1 m-tile fastest
    int stagger;                      // start the second resident block of every CU half a block-time late
    int K;                            // packed K per output channel
    double* stats;                    // training: per (row block, channel) partial (sum, sum^2) of the stored tile, or NULL
    // BatchNorm-backward fusion (data-gradient launches of the fine-tune step): the tile being stored is dL/dy of a
    // Conv+BN(+ReLU) layer whose conv output is bz (same NHWC layout as y).  The epilogue applies that layer's ReLU mask
    // (bmy > 0 if given, else bz*bsc+bbi > 0 if bsc is given, else none), stores g = masked gradient and accumulates the
    // per-channel (sum g, sum g*xhat), xhat = (bz - bmu)*bis, into `stats` — the reduction pass of the BN backward.
    const float* bz;
    const float* bmy;
    const float* bsc;
    const float* bbi;
    const float* bmu;
    const float* bis;
    // dual-source 1x1 (projection shortcut fused into the block's last conv): k-tiles 0..k1-1 read x (C1 = Cin channels,
    // one row per output pixel), k-tiles k1.. read x2 (C2 channels, an H2 x W2 image sampled with stride2)
    const float* x2;
    int k1, C2, H2, W2, stride2;
    unsigned x2_bytes;
    int ablate;                       // profiling only (vatl_tune_set(6, bits), wrong results): 1 = no epilogue
    // opt-in split-K (small batches): blockIdx.z owns k-tiles [z*kt_per_split, ...) and writes a raw partial tile into
    // its slice of `part` (output layout of y, NHWC); splitk_reduce_kernel sums the slices in order and applies the epilogue
    int splits, kt_per_split;
    float* part;
    long long part_slice;
    unsigned x_bytes, w_bytes, y_bytes;   // buffer extents (hardware bounds checks: OOB loads read 0, OOB stores drop)
    FastDivU d_HoWo, d_Wo;                // m -> (image, row, column) without integer divisions (common.h: fdiv; filled in by dispatch())
};

constexpr int BK = 32;
constexpr int LDK = BK;               // LDS row = one k-tile; chunk positions XOR-swizzled (see above)
// dynamic LDS of a BM x BN block: the two k-loop stages, or the epilogue's output tile if that is larger
constexpr int conv_smem_floats(int BM, int BN) { return 2 * (BM + BN) * LDK > BM * (BN + 4) ? 2 * (BM + BN) * LDK : BM * (BN + 4); }
constexpr unsigned OOB = 0xFFFFFFFFu; // byte offset guaranteed outside any descriptor below

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}
__device__ __forceinline__ void buf_store4(__amdgpu_buffer_rsrc_t r, unsigned byte_off, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, byte_off, 0, 0);
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned byte_off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byte_off, 0, 0);
}

// Epilogue shared by the conv kernels: scale/bias in registers, tile staged through LDS, then full-row 16-byte
// stores with the residual read the same way (or per-element stores for NCHW / odd channel counts).
template <int BM, int BN, int WM, int WN, int NT = 256, bool BNB = false>
__device__ __forceinline__ void conv_epilogue(const ConvParams& p, f32x16 (&acc)[WM / 32][WN / 32], float* smem, int m0, int n0,
                                              int ooy, int oox, int wm, int wn, int tid, int lane, int HoWo) {
    constexpr int TM = WM / 32, TN = WN / 32;
    // ---- epilogue -------------------------------------------------------------
    // 1. scale/bias in registers, tile -> LDS (the staging buffers are free after the
    //    loop's last barrier).  D[row = (e&3) + 8*(e>>2) + 4*(lane>>5)][col = lane&31].
    constexpr int LDC = BN + 4;
    float* Cs = smem;
    // 0. output offsets of this thread's float4 columns and the residual tile, requested BEFORE the LDS
    //    transpose so that its HBM latency hides behind the accumulator write-out and the barrier
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.res ? p.y_bytes : 0u, 0x00020000);
    const int OHW = p.OH * p.OW;
    const bool plain = !p.deconv && p.osy == 1 && p.osx == 1 && p.OH == p.Ho && p.OW == p.Wo;   // NHWC output row index == m
    const bool vec = !p.out_nchw && (p.Cout & 3) == 0;
    constexpr int C4 = BN / 4;                         // float4 columns per tile row
    constexpr int RPP = NT / C4;                       // tile rows per pass
    constexpr int NP = BM / RPP;                       // passes
    unsigned offv[NP];
    f32x4 rsv[NP];
    if (vec) {
        const int c4 = tid % C4, r0 = tid / C4;
        const int n = n0 + c4 * 4;
        const bool nv = n < p.Cout;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int m = m0 + r0 + u * RPP;
            int orow = m;
            if (!plain) {
                const int b = fdiv(m, p.d_HoWo);
                const int rem = m - b * HoWo;
                const int oy = fdiv(rem, p.d_Wo);
                const int ox = rem - oy * p.Wo;
                orow = b * OHW + (oy * p.osy + ooy) * p.OW + (ox * p.osx + oox);
            }
            offv[u] = (nv && m < p.M) ? (unsigned)(orow * p.Cout + n) << 2 : OOB;
        }
        if (p.res) {
#pragma unroll
            for (int u = 0; u < NP; ++u) rsv[u] = buf_load4(rr, offv[u]);
        } else {
#pragma unroll
            for (int u = 0; u < NP; ++u) rsv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int cl = wn * WN + j * 32 + (lane & 31);
        const int n = n0 + cl;
        const bool nv = n < p.Cout;
        const float sc = (nv && p.scale) ? p.scale[n] : 1.f;
        const float bi = (nv && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                Cs[row * LDC + cl] = acc[i][j][e] * sc + bi;
            }
    }
    __syncthreads();

    // 2. LDS -> HBM with full rows: (+ residual) (ReLU), branch-free through descriptors
    const float lo = p.relu ? 0.f : -INFINITY;
    if (vec) {
        const int c4 = tid % C4, r0 = tid / C4;
        f32x4 ssum = {0.f, 0.f, 0.f, 0.f}, ssq = {0.f, 0.f, 0.f, 0.f};
        if constexpr (BNB) {
            // BatchNorm-backward fusion (separate instantiations: the extra tile of z / mask registers must not cost the
            // inference kernels their occupancy): mask the gradient tile with the consumer layer's ReLU, store g, reduce (g, g*xhat)
            const __amdgpu_buffer_rsrc_t zr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bz), 0, p.y_bytes, 0x00020000);
            const __amdgpu_buffer_rsrc_t mr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bmy), 0, p.bmy ? p.y_bytes : 0u, 0x00020000);
            const int n = n0 + c4 * 4;
            const bool nv = n < p.Cout;                    // Cout % 4 == 0 on this path: the float4 is in range or entirely out
            const f32x4 one = {1.f, 1.f, 1.f, 1.f}, nul = {0.f, 0.f, 0.f, 0.f};
            const f32x4 mu = nv ? *reinterpret_cast<const f32x4*>(p.bmu + n) : nul, is = nv ? *reinterpret_cast<const f32x4*>(p.bis + n) : nul;
            const f32x4 msc = (nv && p.bsc) ? *reinterpret_cast<const f32x4*>(p.bsc + n) : nul;
            const f32x4 mbi = (nv && p.bsc) ? *reinterpret_cast<const f32x4*>(p.bbi + n) : one;   // no mask: 0*z + 1 > 0
            f32x4 zt[NP], yt[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) zt[u] = buf_load4(zr, offv[u]);
            if (p.bmy) {
#pragma unroll
                for (int u = 0; u < NP; ++u) yt[u] = buf_load4(mr, offv[u]);
            }
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[(r0 + u * RPP) * LDC + c4 * 4]);
                f32x4 g;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float d = v[c] + rsv[u][c];
                    const bool on = p.bmy ? yt[u][c] > 0.f : fmaf(zt[u][c], msc[c], mbi[c]) > 0.f;
                    g[c] = on ? d : 0.f;
                    ssum[c] += g[c];
                    ssq[c] += g[c] * ((zt[u][c] - mu[c]) * is[c]);       // rows >= M: d = 0 exactly
                }
                buf_store4(yr, offv[u], g);
            }
        } else {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&Cs[(r0 + u * RPP) * LDC + c4 * 4]);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = fmaxf(v[c] + rsv[u][c], lo);
            buf_store4(yr, offv[u], o);
            if (p.stats) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { ssum[c] += o[c]; ssq[c] += o[c] * o[c]; }   // rows >= M hold exact zeros
            }
        }
        }
        if (p.stats) {
            // BatchNorm batch statistics of the tile just stored (training forward): per-thread fp32 sums over NP rows,
            // combined over the RPP row groups in double, one (sum, sum^2) pair per (row block, channel) — the layout
            // bn_train_finalize_kernel reduces in a fixed order (deterministic, no atomics).
            lds_barrier();                                 // every thread is done reading Cs (LDS hand-off only: __syncthreads() would wait for the tile's stores)
            f32x4* sh = reinterpret_cast<f32x4*>(smem);
            sh[tid] = ssum; sh[NT + tid] = ssq;
            lds_barrier();
            if (tid < C4) {
                double ds[4] = {0, 0, 0, 0}, dq[4] = {0, 0, 0, 0};
#pragma unroll 2                                           // (full unrolling cost the 128x32 kernel 256 VGPRs and spills)
                for (int k = 0; k < RPP; ++k) {
                    const f32x4 a = sh[k * C4 + tid], b = sh[NT + k * C4 + tid];
#pragma unroll
                    for (int c = 0; c < 4; ++c) { ds[c] += a[c]; dq[c] += b[c]; }
                }
                const long long rb = (long long)blockIdx.y * p.m_tiles + m0 / BM;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const int n = n0 + tid * 4 + c;
                    if (n < p.Cout) {
                        p.stats[(rb * p.Cout + n) * 2 + 0] = ds[c];
                        p.stats[(rb * p.Cout + n) * 2 + 1] = dq[c];
                    }
                }
            }
        }
    } else {
        // NCHW output (heat-map head) or a channel count that is not a multiple of 4:
        // one tile row per thread, lanes run along pixels (contiguous in NCHW)
        constexpr int CPP = NT / BM;                   // tile columns per pass
        const int row = tid % BM, cl0 = tid / BM;
        const int m = m0 + row;
        const bool mv = m < p.M;
        const int b = fdiv(m, p.d_HoWo);
        const int rem = m - b * HoWo;
        const int oy = fdiv(rem, p.d_Wo);
        const int ox = rem - oy * p.Wo;
        const int opix = (oy * p.osy + ooy) * p.OW + (ox * p.osx + oox);
        const int nstride = p.out_nchw ? OHW : 1;
        const int obase = p.out_nchw ? b * p.Cout * OHW + opix : (b * OHW + opix) * p.Cout;
        if (p.res) {
#pragma unroll 4
            for (int ps = 0; ps < BN / CPP; ++ps) {
                const int cl = cl0 + ps * CPP;
                const int n = n0 + cl;
                const unsigned off = (mv && n < p.Cout) ? (unsigned)(obase + n * nstride) << 2 : OOB;
                buf_store1(yr, off, fmaxf(Cs[row * LDC + cl] + buf_load1(rr, off), lo));
            }
        } else {
#pragma unroll 8
            for (int ps = 0; ps < BN / CPP; ++ps) {
                const int cl = cl0 + ps * CPP;
                const int n = n0 + cl;
                const unsigned off = (mv && n < p.Cout) ? (unsigned)(obase + n * nstride) << 2 : OOB;
                buf_store1(yr, off, fmaxf(Cs[row * LDC + cl], lo));
            }
        }
    }
}

// VAR selects the k-loop schedule (tuning knob, see vatl_tune_set).  Measured on MI355X, 3x3 512->512
// @8x6, batch 1024 (tools/conv_bench.py, TFLOP/s): VAR0 126, VAR2 132, VAR3 131, VAR4 138; ablations of
// VAR2: no global loads / LDS writes 150, no barrier 130, no fragment reads 140 (profiles/r01_notes.md).
//   0  two-phase: [address math + global loads] [64 MFMA with fragment reads] [LDS writes] barrier
//   2  fine-grained: the next tile's loads/address math and the next group's fragment reads are
//      issued in the shadow of the MFMAs (one 32x32x2 MFMA occupies the matrix pipe for 64 cycles;
//      the wave is free to issue other instructions meanwhile), pinned with sched_group_barrier
//   4  2 + distance-2 prefetch through two staging register sets (DEFAULT)
//   5  LDS-DMA kernel below (buffer_load ... lds, source-side XOR swizzle): 134 — equal to VAR2, the saved
//      ds_write pass (+5 %, ablation VAR13) is offset by its distance-1 prefetch
// (a rotated loop that buries the tile hand-over in the last MFMA group measured equal to VAR2 and was dropped)
template <int BM, int BN, int WM, int WN, bool STEM, int VAR, bool DUAL = false, int NT = 256, bool BNB = false>
__global__ __launch_bounds__(NT, 2) void conv_igemm_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;           // [2][BN][LDK]

    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int RP = NT / 8;                 // tile rows staged per pass (8 threads x 16 bytes cover the 32-wide k-tile)
    constexpr int LA = BM / RP, LB = BN / RP;  // 16-byte loads per thread per k-tile
    static_assert((BM / WM) * (BN / WN) == NT / 64, "one wave per wave tile");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    // XCD-aware tile order: block b runs on XCD b%8; give each XCD a contiguous
    // run of tiles with the n-tile fastest so the blocks sharing an A panel hit
    // the same L2 (bijective for any grid size).
    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r8 = nblk & 7;
    const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
    int m_tile, n_tile;
    if (p.order == 0) { m_tile = t / p.n_tiles; n_tile = t - m_tile * p.n_tiles; }
    else              { n_tile = t / p.m_tiles; m_tile = t - n_tile * p.m_tiles; }
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    // All resident blocks of a launch start together and take the same time, so chip-wide the matrix-bound
    // k-loops and the memory-bound epilogues would alternate instead of overlapping.  Delaying the second block
    // of every CU (blocks 256..511 under in-order dispatch; later blocks inherit the phase of the slot they
    // take over) by half a k-loop puts the two phases side by side.  Performance only.
    if (p.stagger && bid >= 256 && bid < 512 && nblk >= 1024) {
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    int pad_y = p.pad_y, pad_x = p.pad_x, ooy = p.ooy, oox = p.oox;
    const float* wbase = p.w;
    if (p.deconv) {
        const int py = blockIdx.y >> 1, px = blockIdx.y & 1;
        pad_y = 1 - py; pad_x = 1 - px; ooy = py; oox = px;
        wbase += (long long)blockIdx.y * p.CoutPad * p.K;
    }
    // All global traffic goes through buffer descriptors: 32-bit byte offsets, and an
    // out-of-range offset (OOB) reads zeros / drops the store, so image borders, M
    // tails and the optional residual need no branches (branches made hipcc drain
    // vmcnt before the MFMAs in the first version of this kernel).
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DUAL ? p.x2 : p.x), 0, DUAL ? p.x2_bytes : p.x_bytes, 0x00020000);

    // ---- per-thread gather state --------------------------------------------
    const int lrow = tid >> 3;        // 0..RP-1
    const int kq = tid & 7;           // which float4 of the 32-wide k-tile
    const int wpos = (kq ^ ((lrow >> 1) & 7)) * 4;   // its swizzled position in the LDS row (RP is a multiple of 16: same for every pass)
    int abase[LA], iy0[LA], ix0[LA];  // element offset of (b, iy0, ix0, kq*4); may be negative
    int abase2[DUAL ? LA : 1];        // DUAL: element offset of the row's pixel in the second source
    const int HoWo = p.Ho * p.Wo;
    // 1x1 / stride 1 / pad 0: the input pixel of GEMM row m is pixel m — no (image, row, column) split, i.e. none of the two integer
    // divisions per staged row (8 per thread and block for a 128-row tile)
    const bool flat = !STEM && !DUAL && p.R == 1 && p.S == 1 && p.stride == 1 && pad_y == 0 && pad_x == 0 && p.H == p.Ho && p.W == p.Wo;
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int m = m0 + lrow + RP * i;
        if (flat && m < p.M) {
            iy0[i] = 0; ix0[i] = 0;
            abase[i] = m * p.Cin + kq * 4;
        } else if (m < p.M) {
            const int b = fdiv(m, p.d_HoWo);
            const int rem = m - b * HoWo;
            const int oy = fdiv(rem, p.d_Wo);
            const int ox = rem - oy * p.Wo;
            iy0[i] = oy * p.stride - pad_y;
            ix0[i] = ox * p.stride - pad_x;
            abase[i] = ((b * p.H + iy0[i]) * p.W + ix0[i]) * p.Cin + kq * 4;
            if (DUAL) abase2[i] = ((b * p.H2 + oy * p.stride2) * p.W2 + ox * p.stride2) * p.C2 + kq * 4;
        } else {
            iy0[i] = -(1 << 20); ix0[i] = -(1 << 20); abase[i] = 0;
            if (DUAL) abase2[i] = 0;
        }
    }
    unsigned boff[LB];
#pragma unroll
    for (int j = 0; j < LB; ++j) boff[j] = (unsigned)(((n0 + lrow + RP * j) * p.K + kq * 4) * 4);

    f32x4 ra[LA], rb[LB];
    int kbase = 0;                     // split-K: first k-tile of this block's share
    if (p.splits > 1) {
        kbase = blockIdx.z * p.kt_per_split;
        p.ktiles = min(p.kt_per_split, p.ktiles - kbase);
    }
    int g_r = 0, g_s = 0, g_off = 0;
    bool g_sel = false;                // DUAL: this k-tile reads the second source
    auto gtap = [&](int kt) {          // filter tap / channel offset of k-tile kt (wave-uniform)
        kt += kbase;
        if (STEM) {                    // k-tile = one filter row: 8 taps x 4 channels
            g_r = kt; g_s = kq; g_off = kt * p.W * p.Cin;
        } else if (DUAL) {             // 1x1 over [source 1 channels | source 2 channels]
            g_sel = kt >= p.k1;
            g_r = 0; g_s = 0; g_off = (g_sel ? kt - p.k1 : kt) * BK;
        } else {
            const int rs = kt / p.kpr;
            const int c0 = (kt - rs * p.kpr) * BK;
            g_r = rs / p.S; g_s = rs - g_r * p.S;
            g_off = (g_r * p.W + g_s) * p.Cin + c0;
        }
    };
    auto a_load = [&](int i, bool live) -> f32x4 {     // the A-operand float4 of tile row lrow + 32 i for the current tap
        const bool ok = live && (unsigned)(iy0[i] + g_r) < (unsigned)p.H && (unsigned)(ix0[i] + g_s) < (unsigned)p.W;
        if (DUAL) return buf_load4(g_sel ? xr2 : xr, ok ? (unsigned)((g_sel ? abase2[i] : abase[i]) + g_off) << 2 : OOB);
        return buf_load4(xr, ok ? (unsigned)(abase[i] + g_off) << 2 : OOB);
    };
    auto gloadA = [&](int i, bool live) { ra[i] = a_load(i, live); };
    auto gloadB = [&](int j, int kt, bool live) {
        rb[j] = buf_load4(wr, live ? boff[j] + (unsigned)(kt + kbase) * (BK * 4) : OOB);
    };
    auto gload = [&](int kt, bool live) {
        gtap(kt);
#pragma unroll
        for (int i = 0; i < LA; ++i) gloadA(i, live);
#pragma unroll
        for (int j = 0; j < LB; ++j) gloadB(j, kt, live);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i)
            *reinterpret_cast<f32x4*>(&As[(buf * BM + lrow + RP * i) * LDK + wpos]) = ra[i];
#pragma unroll
        for (int j = 0; j < LB; ++j)
            *reinterpret_cast<f32x4*>(&Bs[(buf * BN + lrow + RP * j) * LDK + wpos]) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing: row frow (+ 32 i), logical chunk 2g + (lane >> 5), stored at chunk ^ ((frow >> 1) & 7)
    const int frow = lane & 31;
    int koff[BK / 8];
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) koff[g] = ((2 * g + (lane >> 5)) ^ ((frow >> 1) & 7)) * 4;

    gload(0, true);
    lstore(0);
    __syncthreads();

    auto frag_read = [&](f32x4 (&af)[TM], f32x4 (&bf)[TN], int buf, int g) {
        const float* Ab = As + (buf * BM + wm * WM + frow) * LDK + koff[g];
        const float* Bb = Bs + (buf * BN + wn * WN + frow) * LDK + koff[g];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK);
    };
    auto mfma_group = [&](const f32x4 (&af)[TM], const f32x4 (&bf)[TN]) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][tt], bf[j][tt], acc[i][j], 0, 0, 0);
    };

    if (VAR == 0) {
        for (int kt = 0; kt < p.ktiles; ++kt) {
            const int buf = kt & 1;
            const bool more = kt + 1 < p.ktiles;
            if (more) gload(kt + 1, true);
#pragma unroll
            for (int g = 0; g < BK / 8; ++g) {
                f32x4 af[TM], bf[TN];
                frag_read(af, bf, buf, g);
                mfma_group(af, bf);
            }
            if (more) lstore(buf ^ 1);
            __syncthreads();
        }
    } else if (VAR == 4) {
        // Distance-2 prefetch: tile kt+2 is requested while tile kt is multiplied and is written to
        // LDS a full k-tile later (two staging register sets, loop unrolled by two so that the sets
        // are statically indexed).  Covers L2-miss latency (MALL/HBM, ~2 us) with >= 4096 pipe cycles.
        constexpr int NG = BK / 8;
        constexpr int MPG = 4 * TM * TN;
        f32x4 sa[2][LA], sb[2][LB];
        auto issue = [&](f32x4 (&da)[LA], f32x4 (&db)[LB], int kt) {      // kt may be past the end: OOB loads
            const bool live = kt < p.ktiles;
            gtap(kt);
#pragma unroll
            for (int i = 0; i < LA; ++i) da[i] = a_load(i, live);
#pragma unroll
            for (int j = 0; j < LB; ++j) db[j] = buf_load4(wr, live ? boff[j] + (unsigned)(kt + kbase) * (BK * 4) : OOB);
        };
        auto stash = [&](const f32x4 (&da)[LA], const f32x4 (&db)[LB], int buf) {
#pragma unroll
            for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4*>(&As[(buf * BM + lrow + RP * i) * LDK + wpos]) = da[i];
#pragma unroll
            for (int j = 0; j < LB; ++j) *reinterpret_cast<f32x4*>(&Bs[(buf * BN + lrow + RP * j) * LDK + wpos]) = db[j];
        };
        issue(sa[0], sb[0], 1);
        for (int kt = 0; kt < p.ktiles; kt += 2) {
            // step kt   : stash set0 (tile kt+1), request tile kt+2 into set1
            // step kt+1 : stash set1 (tile kt+2), request tile kt+3 into set0
            {
                const int buf = kt & 1;
                f32x4 af[2][TM], bf[2][TN];
                frag_read(af[0], bf[0], buf, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 1 < NG) frag_read(af[(g + 1) & 1], bf[(g + 1) & 1], buf, g + 1);
                    if (g == 0) issue(sa[1], sb[1], kt + 2);
                    if (g == NG - 1) stash(sa[0], sb[0], buf ^ 1);
                    mfma_group(af[g & 1], bf[g & 1]);
                    if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
                    for (int q = 0; q < MPG; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x216, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
            }
            if (kt + 1 < p.ktiles) {
                const int buf = (kt + 1) & 1;
                f32x4 af[2][TM], bf[2][TN];
                frag_read(af[0], bf[0], buf, 0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (g + 1 < NG) frag_read(af[(g + 1) & 1], bf[(g + 1) & 1], buf, g + 1);
                    if (g == 0) issue(sa[0], sb[0], kt + 3);
                    if (g == NG - 1) stash(sa[1], sb[1], buf ^ 1);
                    mfma_group(af[g & 1], bf[g & 1]);
                    if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
                    for (int q = 0; q < MPG; ++q) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x216, 2, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
            }
        }
    } else {
        // VAR 2 = fine-grained schedule with distance-1 prefetch.  VAR 10..13 are ABLATIONS for profiling only (wrong results):
        // 10 = no global loads / LDS writes, 11 = no barrier, 12 = no fragment reads in the loop.
#ifdef VATL_ABLATION
        constexpr bool NO_GL = VAR == 10, NO_BAR = VAR == 11, NO_FRAG = VAR == 12, NO_ST = VAR == 13;   // 13 = loads kept, no LDS writes
#else
        static_assert(VAR < 10, "schedule variants 10..13 are profiling ablations: build with -DVATL_ABLATION (build.py --ablation)");
        constexpr bool NO_GL = false, NO_BAR = false, NO_FRAG = false, NO_ST = false;
#endif
        constexpr int NG = BK / 8;                          // 4 fragment groups per k-tile
        constexpr int MPG = 4 * TM * TN;                    // MFMAs per group
        f32x4 af[2][TM], bf[2][TN];
        if (NO_FRAG) { frag_read(af[0], bf[0], 0, 0); frag_read(af[1], bf[1], 0, 1); }
        for (int kt = 0; kt < p.ktiles; ++kt) {
            const int buf = kt & 1;
            const bool live = kt + 1 < p.ktiles;
            if (!NO_FRAG) frag_read(af[0], bf[0], buf, 0);
            gtap(kt + 1);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG && !NO_FRAG) frag_read(af[(g + 1) & 1], bf[(g + 1) & 1], buf, g + 1);
                // the next tile's global loads ride in the first two groups (A then B) so that
                // two groups of MFMAs (>= 2048 pipe cycles) cover their latency before the LDS writes
                if (g == 0 && !NO_GL) {
#pragma unroll
                    for (int i = 0; i < LA; ++i) gloadA(i, live);
                }
                if (g == 1 && !NO_GL) {
#pragma unroll
                    for (int j = 0; j < LB; ++j) gloadB(j, kt + 1, live);
                }
                mfma_group(af[g & 1], bf[g & 1]);
                // pin the interleave: fragment reads first, then MFMAs with the VALU/VMEM work in their shadow
                if (g + 1 < NG && !NO_FRAG) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
                for (int q = 0; q < MPG; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x016, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!NO_GL && !NO_ST) lstore(buf ^ 1);
            if (NO_ST) {                                   // keep the loaded registers alive
#pragma unroll
                for (int i = 0; i < LA; ++i) asm volatile("" ::"v"(ra[i]));
#pragma unroll
                for (int j = 0; j < LB; ++j) asm volatile("" ::"v"(rb[j]));
            }
            if (!NO_BAR) __syncthreads();
        }
        if (NO_BAR) __syncthreads();
    }

    if (p.splits > 1) {                // raw partial sums into this split's slice; scale / bias / residual / ReLU happen in the reduction
        p.y = p.part + (long long)blockIdx.z * p.part_slice;
        p.scale = nullptr; p.bias = nullptr; p.res = nullptr; p.relu = 0; p.out_nchw = 0; p.stats = nullptr;
    }
#ifdef VATL_ABLATION
    if (p.ablate & 1) {                // profiling build only: keep the accumulators alive, skip the write-out
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) s += acc[i][j][e];
        if (s == 12345.678f) p.y[0] = s;
        return;
    }
#endif
    conv_epilogue<BM, BN, WM, WN, NT, BNB>(p, acc, smem, m0, n0, ooy, oox, wm, wn, tid, lane, HoWo);
}

// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// Persistent 1x1 / stride-1 kernel (a plain GEMM Y[M][N] = A[M][K] W[N][K]^T with the conv epilogue).
// The short-K 1x1 layers (conv3 / conv1 of the bottlenecks, K = 64..512) spend as long in the prologue (first
// operand loads with nothing to overlap) and in the tile write-out as in their 2-16 k-tiles (profiles/r01_notes.md,
// ablation knob 6).  Here a block stays resident, walks a run of tiles and requests the NEXT tile's first operand
// tile before the current tile's epilogue, so that latency and the write-out overlap.  Same LDS layout, fragment
// mapping, MFMA order and epilogue as conv_igemm_kernel: results are bit-identical.
// ---------------------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void gemm1x1_persistent_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;           // [2][BN][LDK]
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int LA = BM / 32, LB = BN / 32;
    constexpr int NG = BK / 8, MPG = 4 * TM * TN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lrow = tid >> 3, kq = tid & 7;
    const int wpos = (kq ^ ((lrow >> 1) & 7)) * 4;
    const int frow = lane & 31;
    int koff[BK / 8];
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) koff[g] = ((2 * g + (lane >> 5)) ^ ((frow >> 1) & 7)) * 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

    // XCD-aware runs: XCD x (= block id mod 8) owns tiles [x*per, (x+1)*per) in n-fastest order; its blocks walk the run
    // with a stride of (blocks per XCD), so the blocks that share an activation panel are on the same L2 at the same time
    const int total = p.m_tiles * p.n_tiles;
    const int per = (total + 7) >> 3, bpx = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int run_end = min((xcd + 1) * per, total);
    int t = xcd * per + loc;
    if (t >= run_end) return;

    unsigned aoff[LA], boff[LB];               // byte offsets of this thread's float4s at k-tile 0
    int m0, n0;
    auto setup = [&](int tile, unsigned (&ao)[LA], unsigned (&bo)[LB], int& mm0, int& nn0) {
        const int m_tile = tile / p.n_tiles, n_tile = tile - m_tile * p.n_tiles;
        mm0 = m_tile * BM; nn0 = n_tile * BN;
#pragma unroll
        for (int i = 0; i < LA; ++i) ao[i] = (unsigned)(((mm0 + lrow + 32 * i) * p.K + kq * 4) * 4);   // rows >= M: past the descriptor -> zeros
#pragma unroll
        for (int j = 0; j < LB; ++j) bo[j] = (unsigned)(((nn0 + lrow + 32 * j) * p.K + kq * 4) * 4);
    };
    f32x4 ra[LA], rb[LB];
    auto gload = [&](const unsigned (&ao)[LA], const unsigned (&bo)[LB], int kt, bool live) {
#pragma unroll
        for (int i = 0; i < LA; ++i) ra[i] = buf_load4(xr, live ? ao[i] + (unsigned)kt * (BK * 4) : OOB);
#pragma unroll
        for (int j = 0; j < LB; ++j) rb[j] = buf_load4(wr, live ? bo[j] + (unsigned)kt * (BK * 4) : OOB);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4*>(&As[(buf * BM + lrow + 32 * i) * LDK + wpos]) = ra[i];
#pragma unroll
        for (int j = 0; j < LB; ++j) *reinterpret_cast<f32x4*>(&Bs[(buf * BN + lrow + 32 * j) * LDK + wpos]) = rb[j];
    };
    auto frag_read = [&](f32x4 (&af)[TM], f32x4 (&bf)[TN], int buf, int g) {
        const float* Ab = As + (buf * BM + wm * WM + frow) * LDK + koff[g];
        const float* Bb = Bs + (buf * BN + wn * WN + frow) * LDK + koff[g];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK);
    };

    setup(t, aoff, boff, m0, n0);
    gload(aoff, boff, 0, true);
    lstore(0);
    __syncthreads();
    const int HoWo = p.Ho * p.Wo;
    for (;;) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        f32x4 af[2][TM], bf[2][TN];
        for (int kt = 0; kt < p.ktiles; ++kt) {
            const int buf = kt & 1;
            const bool live = kt + 1 < p.ktiles;
            frag_read(af[0], bf[0], buf, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) frag_read(af[(g + 1) & 1], bf[(g + 1) & 1], buf, g + 1);
                if (g == 0) gload(aoff, boff, kt + 1, live);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][tt], bf[g & 1][j][tt], acc[i][j], 0, 0, 0);
                if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
                for (int q = 0; q < MPG; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x016, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (live) lstore(buf ^ 1);
            __syncthreads();
        }
        // request the next tile's first operand tile: in flight during this tile's epilogue
        const int tn = t + bpx;
        const bool more = tn < run_end;
        unsigned aoffn[LA], boffn[LB];
        int m0n = 0, n0n = 0;
        if (more) { setup(tn, aoffn, boffn, m0n, n0n); gload(aoffn, boffn, 0, true); }
        conv_epilogue<BM, BN, WM, WN>(p, acc, smem, m0, n0, 0, 0, wm, wn, tid, lane, HoWo);
        if (!more) break;
        __syncthreads();                       // every thread is done with the epilogue's LDS tile
        lstore(0);
        __syncthreads();
        t = tn; m0 = m0n; n0 = n0n;
#pragma unroll
        for (int i = 0; i < LA; ++i) aoff[i] = aoffn[i];
#pragma unroll
        for (int j = 0; j < LB; ++j) boff[j] = boffn[j];
    }
}

// The same persistent GEMM with a DISTANCE-2 operand stream that runs across tile boundaries.  With K = 64 .. 256 a tile has only
// 2 .. 8 k-tiles; with one k-tile of look-ahead every one of them waits for an HBM round trip that 4096 matrix-pipe cycles do
// not cover, and the pipe idles (l2.n.c3, K = 128: 61 % of peak at 3.4 TB/s — neither roof).  Here two staging register sets
// alternate: while k-tile kt is multiplied, k-tile kt+1 sits in registers waiting to be written to LDS and k-tile kt+2 is being
// requested — and "kt+2" simply continues into the NEXT tile's first two k-tiles, which therefore are in flight during the
// whole epilogue of the current tile.  Requires an even number of k-tiles (statically indexed register sets).  Same LDS
// layout, fragment mapping, MFMA order and epilogue: bit-identical to the other two kernels.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void gemm1x1_persistent2_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;           // [2][BN][LDK]
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int LA = BM / 32, LB = BN / 32;
    constexpr int NG = BK / 8, MPG = 4 * TM * TN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lrow = tid >> 3, kq = tid & 7;
    const int wpos = (kq ^ ((lrow >> 1) & 7)) * 4;
    const int frow = lane & 31;
    int koff[BK / 8];
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) koff[g] = ((2 * g + (lane >> 5)) ^ ((frow >> 1) & 7)) * 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);

    const int total = p.m_tiles * p.n_tiles;
    const int per = (total + 7) >> 3, bpx = gridDim.x >> 3;
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int run_end = min((xcd + 1) * per, total);
    int t = xcd * per + loc;
    if (t >= run_end) return;

    unsigned aoff[LA], boff[LB];               // byte offsets of this thread's float4s at k-tile 0
    int m0, n0;
    auto setup = [&](int tile, unsigned (&ao)[LA], unsigned (&bo)[LB], int& mm0, int& nn0) {
        const int m_tile = tile / p.n_tiles, n_tile = tile - m_tile * p.n_tiles;
        mm0 = m_tile * BM; nn0 = n_tile * BN;
#pragma unroll
        for (int i = 0; i < LA; ++i) ao[i] = (unsigned)(((mm0 + lrow + 32 * i) * p.K + kq * 4) * 4);   // rows >= M: past the descriptor -> zeros
#pragma unroll
        for (int j = 0; j < LB; ++j) bo[j] = (unsigned)(((nn0 + lrow + 32 * j) * p.K + kq * 4) * 4);
    };
    f32x4 sa[2][LA], sb[2][LB];
    auto issue = [&](f32x4 (&da)[LA], f32x4 (&db)[LB], const unsigned (&ao)[LA], const unsigned (&bo)[LB], int kt, bool live) {
#pragma unroll
        for (int i = 0; i < LA; ++i) da[i] = buf_load4(xr, live ? ao[i] + (unsigned)kt * (BK * 4) : OOB);
#pragma unroll
        for (int j = 0; j < LB; ++j) db[j] = buf_load4(wr, live ? bo[j] + (unsigned)kt * (BK * 4) : OOB);
    };
    auto stash = [&](const f32x4 (&da)[LA], const f32x4 (&db)[LB], int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4*>(&As[(buf * BM + lrow + 32 * i) * LDK + wpos]) = da[i];
#pragma unroll
        for (int j = 0; j < LB; ++j) *reinterpret_cast<f32x4*>(&Bs[(buf * BN + lrow + 32 * j) * LDK + wpos]) = db[j];
    };
    auto frag_read = [&](f32x4 (&af)[TM], f32x4 (&bf)[TN], int buf, int g) {
        const float* Ab = As + (buf * BM + wm * WM + frow) * LDK + koff[g];
        const float* Bb = Bs + (buf * BN + wn * WN + frow) * LDK + koff[g];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK);
    };

    setup(t, aoff, boff, m0, n0);
    issue(sa[0], sb[0], aoff, boff, 0, true);
    stash(sa[0], sb[0], 0);
    issue(sa[0], sb[0], aoff, boff, 1, true);          // ktiles >= 2
    __syncthreads();
    const int HoWo = p.Ho * p.Wo;
    const int KT = p.ktiles;
    for (;;) {
        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        const int tn = t + bpx;
        const bool more = tn < run_end;
        unsigned aoffn[LA], boffn[LB];
        int m0n = 0, n0n = 0;
        setup(more ? tn : t, aoffn, boffn, m0n, n0n);
        f32x4 af[2][TM], bf[2][TN];
        // one k-tile: fragment reads + 64 MFMAs on LDS buffer `buf`, the request of a later k-tile in the first group's shadow,
        // the write of the waiting register set to the other buffer in the last group's
        auto ktile = [&](int buf, auto&& request, auto&& write_back) {
            frag_read(af[0], bf[0], buf, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) frag_read(af[(g + 1) & 1], bf[(g + 1) & 1], buf, g + 1);
                if (g == 0) request();
                if (g == NG - 1) write_back();
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][tt], bf[g & 1][j][tt], acc[i][j], 0, 0, 0);
                if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
                for (int q = 0; q < MPG; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x216, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
        for (int kt = 0; kt < KT; kt += 2) {
            const bool in_tile = kt + 2 < KT;              // k-tiles kt+2, kt+3 belong to this tile; else: the next tile's first two
            ktile(0, [&] { if (in_tile) issue(sa[1], sb[1], aoff, boff, kt + 2, true); else issue(sa[1], sb[1], aoffn, boffn, 0, more); },
                  [&] { stash(sa[0], sb[0], 1); });
            ktile(1, [&] { if (in_tile) issue(sa[0], sb[0], aoff, boff, kt + 3, true); else issue(sa[0], sb[0], aoffn, boffn, 1, more); },
                  [&] { if (in_tile) stash(sa[1], sb[1], 0); });
        }
#ifdef VATL_ABLATION
        if (p.ablate & 1) {                    // profiling build only: keep the accumulators alive, skip the write-out
            float sacc = 0.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) sacc += acc[i][j][e];
            if (sacc == 12345.678f) p.y[0] = sacc;
        } else
#endif
        conv_epilogue<BM, BN, WM, WN>(p, acc, smem, m0, n0, 0, 0, wm, wn, tid, lane, HoWo);
        if (!more) break;
        __syncthreads();                       // every thread is done with the epilogue's LDS tile
        stash(sa[1], sb[1], 0);                // the next tile's k-tile 0 (requested two k-tiles ago); its k-tile 1 waits in set 0
        __syncthreads();
        t = tn; m0 = m0n; n0 = n0n;
#pragma unroll
        for (int i = 0; i < LA; ++i) aoff[i] = aoffn[i];
#pragma unroll
        for (int j = 0; j < LB; ++j) boff[j] = boffn[j];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stream-K scheduling for launches that cannot fill the chip with whole tiles (fine-tune batches: R50 stage 4 at B = 120 is 360
// tiles of 64x128 on 768 resident block slots, FastPose-R152 stage 3 at 384x288 / B = 32 is 432 — 47 % / 56 % of the chip for
// the whole launch).  The launch is exactly as many blocks as there are resident slots; the work is the flat sequence of
// (tile, k-tile) units, cut into equal contiguous shares, so a block computes the tail of one tile, whole tiles, and the head of
// another.  A share that STARTS inside a tile leaves that piece's raw fp32 accumulators in the block's slab of a caller-owned
// workspace and publishes a flag at once (it is the block's first piece); the block that started the tile reaches it as the LAST
// piece of its own share, adds the slabs of the blocks after it — in block order, a fixed order: bitwise reproducible — and runs
// the ordinary epilogue (BatchNorm statistics / BatchNorm-backward fusion included).  (The first version had the roles the other
// way round — the block holding a tile's last k-tile waited, at the START of its share, for a slab its predecessor wrote at the
// END of its own: a chain of waits through all blocks, 1.7x slower than no stream-K at all.)  Every XCD owns a run of whole tiles and splits it among its own blocks, so a tile's pieces share one L2 and
// a block only ever waits, with all of its own work done, for pieces that blocks publish before doing anything else: as long as
// a handful of the launch's blocks are resident the wait ends (blocks without a successor to wait for retire and free their slots).  Visibility does not depend on that placement: slab stores -> s_waitcnt vmcnt(0) ->
// barrier -> one lane's agent-scope release -> flag (relaxed agent store); the reader polls relaxed with s_sleep, then ONE
// agent-scope acquire + barrier, then plain loads (cdna_hip_programming.md section 6, guideline 16).  The flags are never reset:
// every launch carries a new epoch (the workspace is zeroed once, when it is registered).
// The summation order over K differs from the unsplit kernel's (pieces are added per share), so results agree with it to
// fp32 rounding, not bit for bit: only the training paths register a workspace (vatl_set_streamk_workspace_thread).
// ---------------------------------------------------------------------------------------------------------
struct StreamKArgs {
    float* slabs;          // [grid][BM*BN] raw accumulators in register order
    unsigned* flags;       // [grid] last epoch whose slab is complete
    unsigned epoch;
    int tiles;             // m_tiles * n_tiles
};

template <int BM, int BN, int WM, int WN, bool BNB>
__global__ __launch_bounds__(256, 3) void conv_streamk_kernel(ConvParams p, StreamKArgs sk) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                          // [2][BM][LDK]
    float* Bs = smem + 2 * BM * LDK;           // [2][BN][LDK]
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int RP = 32;
    constexpr int LA = BM / RP, LB = BN / RP;
    constexpr int NG = BK / 8, MPG = 4 * TM * TN;
    static_assert((BM / WM) * (BN / WN) == 4, "one wave per wave tile");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int lrow = tid >> 3, kq = tid & 7;
    const int wpos = (kq ^ ((lrow >> 1) & 7)) * 4;
    const int frow = lane & 31;
    int koff[BK / 8];
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) koff[g] = ((2 * g + (lane >> 5)) ^ ((frow >> 1) & 7)) * 4;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    const int HoWo = p.Ho * p.Wo;
    const int KT = p.ktiles;

    // this block's share: XCD x (= block id mod 8) owns tiles [T x / 8, T (x+1) / 8), its blocks cut that run's units evenly
    const int bid = blockIdx.x, xcd = bid & 7, loc = bid >> 3, bpx = gridDim.x >> 3;
    const int t_lo = (int)((long long)sk.tiles * xcd / 8), t_hi = (int)((long long)sk.tiles * (xcd + 1) / 8);
    const long long U = (long long)(t_hi - t_lo) * KT;
    long long u = U * loc / bpx;
    const long long u_end = U * (loc + 1) / bpx, u_begin = u;

    int abase[LA], iy0[LA], ix0[LA];
    unsigned boff[LB];
    f32x4 ra[LA], rb[LB];
    int g_r = 0, g_s = 0, g_off = 0;
    auto gtap = [&](int kt) {
        const int rs = kt / p.kpr;
        const int c0 = (kt - rs * p.kpr) * BK;
        g_r = rs / p.S; g_s = rs - g_r * p.S;
        g_off = (g_r * p.W + g_s) * p.Cin + c0;
    };
    auto gloadA = [&](int i, bool live) {
        const bool ok = live && (unsigned)(iy0[i] + g_r) < (unsigned)p.H && (unsigned)(ix0[i] + g_s) < (unsigned)p.W;
        ra[i] = buf_load4(xr, ok ? (unsigned)(abase[i] + g_off) << 2 : OOB);
    };
    auto gloadB = [&](int j, int kt, bool live) { rb[j] = buf_load4(wr, live ? boff[j] + (unsigned)kt * (BK * 4) : OOB); };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4*>(&As[(buf * BM + lrow + RP * i) * LDK + wpos]) = ra[i];
#pragma unroll
        for (int j = 0; j < LB; ++j) *reinterpret_cast<f32x4*>(&Bs[(buf * BN + lrow + RP * j) * LDK + wpos]) = rb[j];
    };
    auto frag_read = [&](f32x4 (&af)[TM], f32x4 (&bf)[TN], int buf, int g) {
        const float* Ab = As + (buf * BM + wm * WM + frow) * LDK + koff[g];
        const float* Bb = Bs + (buf * BN + wn * WN + frow) * LDK + koff[g];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDK);
    };
    typedef __attribute__((address_space(1))) unsigned gu32;
    f32x4* const slab_mine = reinterpret_cast<f32x4*>(sk.slabs) + (long long)bid * (BM * BN / 4);

    while (u < u_end) {
        const int tl = (int)(u / KT);
        const int kb = (int)(u - (long long)tl * KT);
        const int ke = (int)((long long)kb + (u_end - u) < (long long)KT ? (long long)kb + (u_end - u) : (long long)KT);
        const int t = t_lo + tl;
        const int m_tile = t / p.n_tiles, n_tile = t - m_tile * p.n_tiles;
        const int m0 = m_tile * BM, n0 = n_tile * BN;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int m = m0 + lrow + RP * i;
            if (m < p.M) {
                const int b = fdiv(m, p.d_HoWo);
                const int rem = m - b * HoWo;
                const int oy = fdiv(rem, p.d_Wo);
                const int ox = rem - oy * p.Wo;
                iy0[i] = oy * p.stride - p.pad_y;
                ix0[i] = ox * p.stride - p.pad_x;
                abase[i] = ((b * p.H + iy0[i]) * p.W + ix0[i]) * p.Cin + kq * 4;
            } else {
                iy0[i] = -(1 << 20); ix0[i] = -(1 << 20); abase[i] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) boff[j] = (unsigned)(((n0 + lrow + RP * j) * p.K + kq * 4) * 4);

        f32x16 acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

        gtap(kb);
#pragma unroll
        for (int i = 0; i < LA; ++i) gloadA(i, true);
#pragma unroll
        for (int j = 0; j < LB; ++j) gloadB(j, kb, true);
        lstore(0);
        __syncthreads();
        f32x4 af[2][TM], bf[2][TN];
        for (int kt = kb; kt < ke; ++kt) {
            const int buf = (kt - kb) & 1;
            const bool live = kt + 1 < ke;
            frag_read(af[0], bf[0], buf, 0);
            gtap(kt + 1);
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) frag_read(af[(g + 1) & 1], bf[(g + 1) & 1], buf, g + 1);
                if (g == 0) {
#pragma unroll
                    for (int i = 0; i < LA; ++i) gloadA(i, live);
                }
                if (g == 1) {
#pragma unroll
                    for (int j = 0; j < LB; ++j) gloadB(j, kt + 1, live);
                }
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][tt], bf[g & 1][j][tt], acc[i][j], 0, 0, 0);
                if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
                for (int q = 0; q < MPG; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x016, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (live) lstore(buf ^ 1);
            __syncthreads();
        }

        if (kb > 0) {
            // the share starts inside this tile (always the block's FIRST piece): hand the accumulators to the block that started
            // the tile — published right away, so that block finds them waiting when it gets to the tile at the END of its share
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4)
                        slab_mine[((i * TN + j) * 4 + e4) * 256 + tid] =
                            f32x4{acc[i][j][4 * e4], acc[i][j][4 * e4 + 1], acc[i][j][4 * e4 + 2], acc[i][j][4 * e4 + 3]};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store((gu32*)(sk.flags + bid), sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else {
            if (ke < KT) {
                // this block started the tile and its share ends inside it (always the block's LAST piece): add the pieces of the
                // blocks after it (same XCD: ids bid + 8, bid + 16, ...) in order, up to the one that holds the tile's last k-tile
                const long long tile_end = (long long)(tl + 1) * KT;
                long long s1 = u_end;
                int q = loc + 1;
                while (s1 < tile_end) {
                    const long long qe = U * (q + 1) / bpx;
                    if (qe == s1) { ++q; continue; }               // (a block without units publishes nothing)
                    const int pb = q * 8 + xcd;
                    if (tid == 0) {
                        while (__hip_atomic_load((gu32*)(sk.flags + pb), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch)
                            __builtin_amdgcn_s_sleep(8);
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    const f32x4* slab = reinterpret_cast<const f32x4*>(sk.slabs) + (long long)pb * (BM * BN / 4);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) {
                                const f32x4 v = slab[((i * TN + j) * 4 + e4) * 256 + tid];
#pragma unroll
                                for (int c = 0; c < 4; ++c) acc[i][j][4 * e4 + c] += v[c];
                            }
                    s1 = qe;
                    ++q;
                }
            }
            conv_epilogue<BM, BN, WM, WN, 256, BNB>(p, acc, smem, m0, n0, p.ooy, p.oox, wm, wn, tid, lane, HoWo);
            __syncthreads();                       // the epilogue's LDS tile is free again
        }
        u += ke - kb;
    }
}

// LDS-DMA variant (VAR 5): operand tiles go HBM/L2 -> LDS directly (buffer_load ... lds, 1 KiB per wave
// instruction, no staging VGPRs, no ds_write pass).  The DMA destination is lane-linear (base + lane*16 B), so
// the LDS rows are unpadded 32-float rows and bank conflicts are avoided by an XOR swizzle applied on the
// SOURCE side: LDS chunk position p of tile row r holds global chunk p ^ ((r>>1)&7); the fragment reader asks
// for chunk (2g+half) ^ ((r>>1)&7).  16 consecutive rows then cover all 16 16-byte slots of a 256-byte bank row
// (conflict-free ds_read_b128).  Out-of-image taps / tail tiles use out-of-range offsets: the DMA writes zeros
// (verified on MI355X).  Two stages of 32 KiB; the epilogue reuses the space.
// ---------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;

// (body in a __device__ function: with the DMA builtin called from a lambda directly inside the __global__
// template, hipcc 7.2 silently drops the HOST stub of the kernel and the library fails to load)
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_igemm_dma_body(const ConvParams& p, float* smem) {
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    constexpr int STAGE = (BM + BN) * BK;            // floats per stage: A rows, then B rows, 32 floats each
    constexpr int IA = BM / 32, IB = BN / 32;        // DMA instructions per wave per stage (8 rows x 128 B each)
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;

    const int nblk = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, loc = bid >> 3, q = nblk >> 3, r8 = nblk & 7;
    const int t = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + loc;
    const int m_tile = t / p.n_tiles;
    const int n_tile = t - m_tile * p.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    int pad_y = p.pad_y, pad_x = p.pad_x, ooy = p.ooy, oox = p.oox;
    const float* wbase = p.w;
    if (p.deconv) {
        const int py = blockIdx.y >> 1, px = blockIdx.y & 1;
        pad_y = 1 - py; pad_x = 1 - px; ooy = py; oox = px;
        wbase += (long long)blockIdx.y * p.CoutPad * p.K;
    }
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wbase), 0, p.w_bytes, 0x00020000);

    // ---- loader state: instruction i of wave w fills tile rows (4i + w)*8 .. +7; lane -> (row, chunk position)
    const int lrow = lane >> 3;                       // row inside the 8-row group
    const int lpos = lane & 7;                        // 16-byte position inside the LDS row
    const int lswz = ((lane >> 4) + 4 * (wave & 1)) & 7;   // ((row >> 1) & 7) for every row this lane loads
    const int lchk = lpos ^ lswz;                     // global 16-byte chunk that belongs at this position
    int abase[IA], iy0[IA], ix0[IA];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int i = 0; i < IA; ++i) {
        const int m = m0 + (4 * i + wave) * 8 + lrow;
        if (m < p.M) {
            const int b = fdiv(m, p.d_HoWo);
            const int rem = m - b * HoWo;
            const int oy = fdiv(rem, p.d_Wo);
            const int ox = rem - oy * p.Wo;
            iy0[i] = oy * p.stride - pad_y;
            ix0[i] = ox * p.stride - pad_x;
            abase[i] = ((b * p.H + iy0[i]) * p.W + ix0[i]) * p.Cin + lchk * 4;
        } else {
            iy0[i] = -(1 << 20); ix0[i] = -(1 << 20); abase[i] = 0;
        }
    }
    unsigned boff[IB];
#pragma unroll
    for (int j = 0; j < IB; ++j) boff[j] = (unsigned)(((n0 + (4 * j + wave) * 8 + lrow) * p.K + lchk * 4) * 4);

    auto issue = [&](int buf, int kt) {               // DMA of k-tile kt into stage buf (zeros past the end)
        const bool live = kt < p.ktiles;
        const int rs = kt / p.kpr;
        const int c0 = (kt - rs * p.kpr) * BK;
        const int r = rs / p.S, sx = rs - r * p.S;
        const int off = (r * p.W + sx) * p.Cin + c0;
        float* stage = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < IA; ++i) {
            const bool ok = live && (unsigned)(iy0[i] + r) < (unsigned)p.H && (unsigned)(ix0[i] + sx) < (unsigned)p.W;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lds_void*)(stage + (4 * i + wave) * 8 * BK), 16,
                                                     ok ? (unsigned)(abase[i] + off) << 2 : OOB, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < IB; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (lds_void*)(stage + (BM + (4 * j + wave) * 8) * BK), 16,
                                                     live ? boff[j] + (unsigned)kt * (BK * 4) : OOB, 0, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // fragment addressing: row = frow (+32 i), logical chunk 2g + half, stored at position chunk ^ ((frow>>1)&7)
    const int frow = lane & 31;
    const int half = lane >> 5;
    const int fswz = (frow >> 1) & 7;
    int koff[BK / 8];
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) koff[g] = ((2 * g + half) ^ fswz) * 4;
    auto frag_read = [&](f32x4 (&af)[TM], f32x4 (&bf)[TN], int buf, int g) {
        const float* Ab = smem + buf * STAGE + (wm * WM + frow) * BK + koff[g];
        const float* Bb = smem + buf * STAGE + (BM + wn * WN + frow) * BK + koff[g];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * BK);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * BK);
    };

    issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    constexpr int NG = BK / 8;
    constexpr int MPG = 4 * TM * TN;
    for (int kt = 0; kt < p.ktiles; ++kt) {
        const int buf = kt & 1;
        f32x4 af[2][TM], bf[2][TN];
        frag_read(af[0], bf[0], buf, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) frag_read(af[(g + 1) & 1], bf[(g + 1) & 1], buf, g + 1);
            if (g == 0) issue(buf ^ 1, kt + 1);
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[g & 1][i][tt], bf[g & 1][j][tt], acc[i][j], 0, 0, 0);
            if (g + 1 < NG) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);
#pragma unroll
            for (int q2 = 0; q2 < MPG; ++q2) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x016, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the next stage has landed in LDS
        __syncthreads();
    }
    conv_epilogue<BM, BN, WM, WN>(p, acc, smem, m0, n0, ooy, oox, wm, wn, tid, lane, HoWo);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256, 2) void conv_igemm_dma_kernel(ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_igemm_dma_body<BM, BN, WM, WN>(p, smem);
}

template <int BM, int BN, int WM, int WN>
static int launch_dma(const ConvParams& p, int phases, hipStream_t st);

static std::atomic<int> g_var{4};      // k-loop schedule (vatl_tune_set(0, v)); 4 = shipped default
static std::atomic<int> g_order{0};    // tile order (vatl_tune_set(1, v))
static std::atomic<int> g_ablate{0};   // vatl_tune_set(6, bits): 1 = no epilogue, 2 = one k-tile only (profiling ablations, wrong results)
static std::atomic<int> g_bm{0};       // tile rows (vatl_tune_set(5, v)): 0 = by grid size, 64 or 128 = forced
static std::atomic<int> g_stagger{0};  // block stagger in percent of the k-loop time (vatl_tune_set(2, v)); 0 = off
static std::atomic<int> g_splitk_policy{0};  // vatl_tune_set(9, v): 0 = cut by the launch's own block count, 1 = batch-invariant cut (per-image geometry)

// Split-K (opt-in, vatl_set_splitk_workspace): y = act(sum_z part[z] * scale + bias (+ residual)), slices summed in order.
// One thread per four channels of one output pixel (Cout % 4 == 0) or per element.
__global__ void splitk_reduce_kernel(const float* __restrict__ part, long long slice, int splits, const float* __restrict__ scale,
                                     const float* __restrict__ bias, const float* __restrict__ res, float* __restrict__ y, long long total, int Cout,
                                     int OHW, int relu, int out_nchw) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cout);
        float v = 0.f;
        for (int z = 0; z < splits; ++z) v += part[(long long)z * slice + i];
        v = v * (scale ? scale[c] : 1.f) + (bias ? bias[c] : 0.f);
        long long o = i;
        if (out_nchw) {
            const long long pix = i / Cout;
            const long long b = pix / OHW;
            o = (b * Cout + c) * OHW + (pix - b * OHW);
        }
        if (res) v += res[o];
        y[o] = relu ? fmaxf(v, 0.f) : v;
    }
}

// caller-owned split-K workspaces (vatl_set_splitk_workspace), one slot per device: a launch only ever uses the workspace that
// was registered while its own device was current.  All NULL = split-K off (g_splitk_any skips the device query).
constexpr int kMaxDevices = 16;
static std::atomic<float*> g_splitk_ws[kMaxDevices];
static std::atomic<long long> g_splitk_floats[kMaxDevices];
static std::atomic<int> g_splitk_any{0};

// ... and the calling host thread's own workspace (vatl_set_splitk_workspace_thread): what the launches of THIS thread use while it
// is set, with the batch-invariant cut policy — several host threads (DataParallel replica threads, one stream each) can run
// small-batch calls side by side without sharing partial-sum buffers or any global switch.
static thread_local float* tl_splitk_ws = nullptr;
static thread_local long long tl_splitk_floats = 0;

static float* splitk_workspace(long long* floats, int* policy) {
    if (tl_splitk_ws) { *floats = tl_splitk_floats; *policy = 1; return tl_splitk_ws; }
    *policy = g_splitk_policy.load(std::memory_order_relaxed);
    if (!g_splitk_any.load(std::memory_order_acquire)) return nullptr;
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return nullptr;
    *floats = g_splitk_floats[d].load(std::memory_order_relaxed);
    return g_splitk_ws[d].load(std::memory_order_acquire);
}

template <int BM, int BN, int WM, int WN, bool STEM, int VAR, bool DUAL = false, int NT = 256, bool BNB = false>
static int launch(const ConvParams& p, int phases, hipStream_t st) {
    auto kern = conv_igemm_kernel<BM, BN, WM, WN, STEM, VAR, DUAL, NT, BNB>;
    constexpr int smem = conv_smem_floats(BM, BN) * (int)sizeof(float);
    static std::atomic<unsigned> configured{0};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, configured, "conv_igemm")) return rc;
    ConvParams q = p;
    q.n_tiles = p.CoutPad / BN;
    q.m_tiles = cdiv(p.M, BM);
    q.order = g_order.load(std::memory_order_relaxed);
#ifdef VATL_ABLATION
    q.ablate = g_ablate.load(std::memory_order_relaxed);
    if ((q.ablate & 2) && q.ktiles > 1) q.ktiles = 1;
#else
    q.ablate = 0;
#endif
    // one s_sleep(127) = 8128 cycles; a k-tile costs ~8192 cycles when two blocks share the SIMDs
    q.stagger = (int)((long long)g_stagger.load(std::memory_order_relaxed) * p.ktiles / 100);
    const int m_tiles = cdiv(p.M, BM);
    const long long blocks = (long long)m_tiles * q.n_tiles * phases;
    long long ws_floats = 0;
    int sk_policy = 0;
    float* ws = splitk_workspace(&ws_floats, &sk_policy);
    q.splits = 1;
    if (ws && !STEM && !DUAL && !q.stats && !q.ablate && q.ktiles >= 16) {
        const long long out_elems = (long long)q.y_bytes / 4;
        int splits = 0;
        if (sk_policy == 1) {
            // batch-invariant policy (module calls with <= 16 crops, vatl_tune_set(9, 1)): the cut depends on the layer's
            // per-image geometry only — block count of a nominal 4-crop batch in 64-row tiles, workspace need of a 16-crop
            // batch — so a crop's bits do not depend on how many crops share its call
            const long long per_img_rows = p.M / (p.N > 0 ? p.N : 1), per_img_out = out_elems / (p.N > 0 ? p.N : 1);
            const long long blocks4 = (long long)cdiv(4 * per_img_rows, 64) * q.n_tiles * phases;
            if (blocks4 < 512) {
                splits = (int)std::min<long long>(q.ktiles / 8, (1024 + blocks4 - 1) / blocks4);
                const long long fit16 = ws_floats / (per_img_out > 0 ? 16 * per_img_out : 1);
                if (splits > fit16) splits = (int)fit16;
                if ((long long)splits * out_elems > ws_floats) splits = 0;          // more than 16 crops under this policy: unsplit
            }
        } else if (blocks < 256) {
            // small launch with a long reduction (single-frame / small-batch inference): cut K so that ~512 blocks are in flight
            splits = (int)std::min<long long>(q.ktiles / 8, (512 + blocks - 1) / blocks);
            const long long fit = ws_floats / (out_elems > 0 ? out_elems : 1);
            if (splits > fit) splits = (int)fit;
        }
        if (splits >= 2) {
            q.kt_per_split = (q.ktiles + splits - 1) / splits;
            q.splits = (q.ktiles + q.kt_per_split - 1) / q.kt_per_split;
            q.part = ws;
            q.part_slice = out_elems;
        }
    }
    dim3 grid((unsigned)(m_tiles * q.n_tiles), (unsigned)phases, (unsigned)q.splits);
    hipLaunchKernelGGL(kern, grid, dim3(NT), smem, st, q);
    meter_add(0, 2.0 * ((double)m_tiles * BM) * ((double)q.n_tiles * BN) * ((double)q.ktiles * BK) * phases);
    meter_route(BNB ? kRouteIgemmBnBwd : kRouteIgemm);
    if (q.splits > 1) {
        const long long total = q.part_slice;
        long long gsz = (total + 255) / 256; if (gsz > 8192) gsz = 8192;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)gsz), dim3(256), 0, st, q.part, q.part_slice, q.splits, p.scale, p.bias, p.res, p.y, total,
                           p.Cout, p.OH * p.OW, p.relu, p.out_nchw);
    }
    return check_launch("conv_igemm");
}

template <int BM, int BN, int WM, int WN>
static int launch_dma(const ConvParams& p, int phases, hipStream_t st) {
    auto kern = conv_igemm_dma_kernel<BM, BN, WM, WN>;
    constexpr int stage_bytes = 2 * (BM + BN) * BK * (int)sizeof(float);
    constexpr int epi_bytes = BM * (BN + 4) * (int)sizeof(float);
    constexpr int smem = stage_bytes > epi_bytes ? stage_bytes : epi_bytes;
    static std::atomic<unsigned> configured{0};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, configured, "conv_igemm_dma")) return rc;
    ConvParams q = p;
    q.n_tiles = p.CoutPad / BN;
    q.m_tiles = cdiv(p.M, BM);
    q.order = 0; q.stagger = 0;
    dim3 grid((unsigned)(q.m_tiles * q.n_tiles), (unsigned)phases, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, q);
    meter_add(0, 2.0 * ((double)q.m_tiles * BM) * ((double)q.n_tiles * BN) * ((double)q.ktiles * BK) * phases);
    meter_route(kRouteIgemmDma);
    return check_launch("conv_igemm_dma");
}

static std::atomic<int> g_persist_dist{2};   // vatl_tune_set(10, v): operand look-ahead of the persistent 1x1 kernel (1 or 2 k-tiles)

template <int BM, int BN, int WM, int WN, bool D2>
static int launch_persistent_impl(const ConvParams& p, hipStream_t st) {
    auto kern = gemm1x1_persistent_kernel<BM, BN, WM, WN>;
    auto kern2 = gemm1x1_persistent2_kernel<BM, BN, WM, WN>;
    constexpr int smem = conv_smem_floats(BM, BN) * (int)sizeof(float);
    static std::atomic<unsigned> configured{0};
    if (int rc = ensure_dynamic_lds(D2 ? reinterpret_cast<const void*>(kern2) : reinterpret_cast<const void*>(kern), smem, configured, "gemm1x1_persistent")) return rc;
    ConvParams q = p;
    q.n_tiles = p.CoutPad / BN;
    q.m_tiles = cdiv(p.M, BM);
    const int total = q.m_tiles * q.n_tiles;
    int grid = 512;                            // two resident blocks per CU
    if (grid > total) grid = (total + 7) / 8 * 8;
#ifdef VATL_ABLATION
    q.ablate = g_ablate.load(std::memory_order_relaxed);
    if (q.ablate & 4) q.y_bytes = 0;           // every output store (and residual load) out of range: dropped, no HBM writes
    if (q.ablate & 8) q.x_bytes = 0;           // every activation load out of range: zeros, no HBM reads
    if ((q.ablate & 2) && q.ktiles > 2) { q.ktiles = 2; }
#endif
    if (D2) hipLaunchKernelGGL(kern2, dim3((unsigned)grid), dim3(256), smem, st, q);
    else    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, st, q);
    meter_add(0, 2.0 * ((double)q.m_tiles * BM) * ((double)q.n_tiles * BN) * ((double)q.ktiles * BK));
    meter_route(kRoutePersistent1x1);
    return check_launch("gemm1x1_persistent");
}

template <int BM, int BN, int WM, int WN>
static int launch_persistent(const ConvParams& p, hipStream_t st) {
    const bool d2 = g_persist_dist.load(std::memory_order_relaxed) == 2 && p.ktiles >= 2 && (p.ktiles & 1) == 0;
    return d2 ? launch_persistent_impl<BM, BN, WM, WN, true>(p, st) : launch_persistent_impl<BM, BN, WM, WN, false>(p, st);
}

static std::atomic<int> g_persist{1};  // vatl_tune_set(7, v): persistent kernel for 1x1 layers with K <= 256 v (0 = off)

// Stream-K workspace of the calling host thread (vatl_set_streamk_workspace_thread): [1024 flag words][kStreamKGrid slabs of 64 x 128
// floats].  Thread-local like the split-K one: replica threads never share slabs, and a thread that registered nothing (every
// inference path) never takes this route.
constexpr int kStreamKGrid = 768;                 // three resident 64x128 blocks on each of the 256 CUs
constexpr long long kStreamKBytes = 4096 + (long long)kStreamKGrid * 64 * 128 * 4;
static thread_local char* tl_streamk_ws = nullptr;
static thread_local unsigned tl_streamk_epoch = 0;
static std::atomic<int> g_streamk{1};             // vatl_tune_set(12, v): 0 = never take the stream-K route

template <bool BNB>
static int launch_streamk(const ConvParams& p, hipStream_t st) {
    auto kern = conv_streamk_kernel<64, 128, 32, 64, BNB>;
    constexpr int smem = conv_smem_floats(64, 128) * (int)sizeof(float);
    static std::atomic<unsigned> configured{0};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, configured, "conv_streamk")) return rc;
    ConvParams q = p;
    q.n_tiles = p.CoutPad / 128;
    q.m_tiles = cdiv(p.M, 64);
    q.splits = 1;
    StreamKArgs sk{};
    sk.flags = reinterpret_cast<unsigned*>(tl_streamk_ws);
    sk.slabs = reinterpret_cast<float*>(tl_streamk_ws + 4096);
    sk.epoch = ++tl_streamk_epoch;
    if (sk.epoch == 0) sk.epoch = ++tl_streamk_epoch;          // 0 is the "never written" value of a fresh workspace
    sk.tiles = q.m_tiles * q.n_tiles;
    hipLaunchKernelGGL(kern, dim3(kStreamKGrid), dim3(256), smem, st, q, sk);
    meter_add(0, 2.0 * ((double)q.m_tiles * 64) * ((double)q.n_tiles * 128) * ((double)q.ktiles * BK));
    meter_route(kRouteStreamK);
    return check_launch("conv_streamk");
}

// Whole-tile launches that leave a large part of the chip idle take the stream-K route (64x128 tiles only: the tile the
// dispatcher picks for small launches): fewer than 85 % of the block slots busy over the launch's rounds, a reduction long
// enough to cut (>= 8 k-tiles) and at least 6 k-tiles of work per block.
static bool streamk_wanted(const ConvParams& p, int phases, int bn, int bm, bool stem) {
    if (!tl_streamk_ws || !g_streamk.load(std::memory_order_relaxed) || stem || phases != 1 || bn != 128 || bm != 64 || p.x2 || p.out_nchw || p.deconv ||
        (p.Cout & 3) || p.ktiles < 8)
        return false;
    const long long tiles = (long long)cdiv(p.M, 64) * (p.CoutPad / 128);
    const long long rounds = (tiles + kStreamKGrid - 1) / kStreamKGrid;
    if (tiles * 100 >= rounds * kStreamKGrid * 85) return false;
    return tiles * p.ktiles >= 6LL * kStreamKGrid && tiles >= 64;
}

// CoutPad granularity the packer must honour for a given Cout.
static int tile_n_for(int Cout) { return Cout <= 32 ? 32 : (Cout <= 64 ? 64 : 128); }

// Rows per block tile (measured on MI355X, tools/conv_bench.py --bm 128,64 at batches 120..1024, profiles/r01_notes.md):
//  * 64-channel outputs: the 64x64 tile (36.8 KB of LDS, four blocks = 16 waves per CU) beats 128x64 (two blocks) at
//    every size (+10..20 %);
//  * 128-wide tiles: 128 rows unless the launch has fewer than ~1.5 rounds of the 512 resident blocks (small batches:
//    the B = 120 fine-tune step, single-frame inference), where 64-row tiles fill the chip better (+15..30 %).
static int tile_m_for(const ConvParams& p, int phases, int bn, bool stem) {
    if (stem || bn < 64) return 128;                      // (a 256x32 tile measured 8 % slower than 128x32 on the HRNet 32-channel branch)
    const int forced = g_bm.load(std::memory_order_relaxed);
    if (forced == 64 || forced == 128) return forced;
    const long long blocks128 = (long long)cdiv(p.M, 128) * (p.CoutPad / bn) * phases;
    // 64-channel outputs: with the swizzled LDS rows the 128x64 tile has three resident blocks per CU and beats the 64x64 tile
    // (four blocks) by 2.5-3 % once the launch has a few rounds of them (l1.c2 1873 -> 1827 us, hr.b64 487 -> 474 us at 1024 crops)
    if (bn == 64) return blocks128 >= 1536 ? 128 : 64;
    return blocks128 < 768 ? 64 : 128;
}

static int dispatch(const ConvParams& p_in, int phases, bool stem, hipStream_t st, int64_t* row_blocks = nullptr) {
    ConvParams p = p_in;
    p.d_HoWo = make_fastdiv((unsigned)(p.Ho * p.Wo)); p.d_Wo = make_fastdiv((unsigned)p.Wo);
    const int bn = tile_n_for(p.Cout);
    if (p.CoutPad % bn != 0) return fail(VATL_EINVAL, "CoutPad %d must be a multiple of %d for Cout %d", p.CoutPad, bn, p.Cout);
    const int var = g_var.load(std::memory_order_relaxed);
    const bool stem64 = stem && bn == 64 && var != 0;     // the 7x7 stem runs the 64x64 tile (see below)
    const int bm = stem64 ? 64 : tile_m_for(p, phases, bn, stem);
    if (row_blocks) *row_blocks = (int64_t)cdiv(p.M, bm) * phases;   // what the statistics epilogue of the chosen tile writes
    if (stem) {
        // 7x7 stem, 1024 crops (tools/conv_bench.py): 128x64 two-phase 3.65 ms, 128x64 prefetch 3.40 ms, 64x64 prefetch 3.06 ms
        if (stem64) return launch<64, 64, 32, 32, true, 4>(p, phases, st);
        if (bn == 64) return launch<128, 64, 64, 32, true, 0>(p, phases, st);
        if (bn == 128) return launch<128, 128, 64, 64, true, 0>(p, phases, st);
        return launch<128, 32, 32, 32, true, 0>(p, phases, st);
    }
    if (streamk_wanted(p, phases, bn, bm, stem)) {
        if (p.bz) {
            if (!p.stats || p.out_nchw) return fail(VATL_EINVAL, "conv bn-backward fusion: needs a statistics buffer and an NHWC output with Cout %% 4 == 0");
            return launch_streamk<true>(p, st);
        }
        return launch_streamk<false>(p, st);
    }
    if (p.bz) {
        // data-gradient launch that also runs the reduction pass of the consumer layer's BatchNorm backward (fine-tune step)
        if (!p.stats || (p.Cout & 3) || p.out_nchw) return fail(VATL_EINVAL, "conv bn-backward fusion: needs a statistics buffer and an NHWC output with Cout %% 4 == 0");
        if (bn == 32) return launch<128, 32, 32, 32, false, 4, false, 256, true>(p, phases, st);
        if (bn == 64 && bm == 64) return launch<64, 64, 32, 32, false, 4, false, 256, true>(p, phases, st);
        if (bn == 64) return launch<128, 64, 64, 32, false, 4, false, 256, true>(p, phases, st);
        if (bm == 64) return launch<64, 128, 32, 64, false, 4, false, 256, true>(p, phases, st);
        return launch<128, 128, 64, 64, false, 4, false, 256, true>(p, phases, st);
    }
    if (bm == 64) {
        if (bn == 128) return launch<64, 128, 32, 64, false, 4>(p, phases, st);
        return launch<64, 64, 32, 32, false, 4>(p, phases, st);
    }
    // short-K 1x1 / stride-1 layers on whole 128x128 tiles: the persistent GEMM kernel
    const int pk = g_persist.load(std::memory_order_relaxed);
    if (pk && bn == 128 && var == 4 && phases == 1 && p.R == 1 && p.S == 1 && p.stride == 1 && p.pad_y == 0 && !p.out_nchw && !p.deconv &&
        p.osy == 1 && p.osx == 1 && p.OH == p.Ho && p.OW == p.Wo && (p.Cout & 3) == 0 && p.ktiles <= pk * 8 && !p.x2 &&
        (long long)(p.M + 128) * p.K < (1LL << 30))
        return launch_persistent<128, 128, 64, 64>(p, st);
    if (bn == 128 && var == 5) return launch_dma<128, 128, 64, 64>(p, phases, st);
    if (bn == 64 && var == 5) return launch_dma<128, 64, 64, 32>(p, phases, st);
    if (bn == 128) {
        if (var == 2) return launch<128, 128, 64, 64, false, 2>(p, phases, st);
        if (var == 4) return launch<128, 128, 64, 64, false, 4>(p, phases, st);
#ifdef VATL_ABLATION
        if (var == 10) return launch<128, 128, 64, 64, false, 10>(p, phases, st);
        if (var == 11) return launch<128, 128, 64, 64, false, 11>(p, phases, st);
        if (var == 12) return launch<128, 128, 64, 64, false, 12>(p, phases, st);
        if (var == 13) return launch<128, 128, 64, 64, false, 13>(p, phases, st);
#endif
        if (var == 0) return launch<128, 128, 64, 64, false, 0>(p, phases, st);
        return launch<128, 128, 64, 64, false, 4>(p, phases, st);
    }
    if (bn == 64) {
        // (a 256x64 tile — two 128-row tiles on one filter tile, 64x64 wave tiles, 238 VGPRs, two blocks per CU — measured 4 % SLOWER than
        // 128x64 with three: hr.b64 480 -> 502 us, l1.c2 1838 -> 1904 us at 1024 crops; not kept)
        if (var == 2) return launch<128, 64, 64, 32, false, 2>(p, phases, st);
        if (var == 0) return launch<128, 64, 64, 32, false, 0>(p, phases, st);
        return launch<128, 64, 64, 32, false, 4>(p, phases, st);
    }
    if (var == 2) return launch<128, 32, 32, 32, false, 2>(p, phases, st);
    if (var == 0 || p.ktiles < 4) return launch<128, 32, 32, 32, false, 0>(p, phases, st);   // K <= 96: nothing to pipeline
    return launch<128, 32, 32, 32, false, 4>(p, phases, st);
}

}  // namespace vatl

namespace vatl {
int conv3x3_halo_try(const float* x, const float* w, const float* scale, const float* bias, const float* residual, float* y, int N, int H, int W,
                     int Cin, int Cout, int CoutPad, int R, int S, int stride, int pad, int relu, hipStream_t st);   // conv3x3_halo.hip
int conv3x3_halo_enable(int on);
int wino_set_ablate(int bits);                                 // conv_winograd.hip
int wino_set_group_kb(int v);
int wino_set_halves(int v);
int wino_set_persist(int v);
int wino_wgrad_set_halves(int v);
int wino_wgrad_set_table(int v);
int wino_set_persist_pf(int v);
int wino_wgrad_set_blocks(int v);                              // winograd_wgrad.hip

}

using namespace vatl;

namespace vatl {                                              // internal hooks behind vatl_tune_set (not part of the C ABI: hidden, C++ linkage)
__attribute__((visibility("hidden"))) int tune_wgrad_blocks(int blocks);   // conv_wgrad.hip
__attribute__((visibility("hidden"))) int crop_tune_px(int px);            // crop.hip
}

extern "C" int vatl_tune_set(int knob, int value) {
    // PRODUCT KNOBS — process-global route selectors (relaxed atomics; set them before launching from several threads).  Every accepted
    // value computes the SAME BITS as the default (tests/test_gpu_conv.py, tests/test_gpu_winograd.py assert it per knob); the table in
    // include/vatl_hip.h is the contract.  Nothing else is accepted by the shipped library.
    switch (knob) {
    case 0:  if (value == 0 || value == 2 || value == 4 || value == 5) { g_var.store(value, std::memory_order_relaxed); return 0; } break;
    case 1:  if (value == 0 || value == 1) { g_order.store(value, std::memory_order_relaxed); return 0; } break;
    case 5:  if (value == 0 || value == 64 || value == 128) { g_bm.store(value, std::memory_order_relaxed); return 0; } break;
    case 7:  if (value >= 0 && value <= 64) { g_persist.store(value, std::memory_order_relaxed); return 0; } break;
    case 8:  if (value == 0 || value == 1) return conv3x3_halo_enable(value); break;
    case 10: if (value == 1 || value == 2) { g_persist_dist.store(value, std::memory_order_relaxed); return 0; } break;
    case 18: if (value >= 0 && value <= (1 << 20)) return wino_set_group_kb(value); break;
    case 21: if (value >= 1 && value <= 3) return wino_set_halves(value); break;
    case 22: if (value >= 0 && value <= 4096) return wino_set_persist(value); break;
    case 24: if (value >= 0 && value <= 3) return wino_set_persist_pf(value); break;
    case 25: if (value == 0 || value == 1) return wino_wgrad_set_table(value); break;
    default: break;
    }
#ifdef VATL_ABLATION
    // PROFILING VARIANT ONLY (build.py --ablation -> libvatl_hip_ablation.so): knobs that change the summation order (3, 9, 12, 19) or are not pinned bit-identical (23), performance-only
    // experiments (2, 16) and the ablations that compute WRONG results by construction (0: 10..13, 4, 6, 17; these also need VATL_ALLOW_ABLATION=1).
    const bool wrong = (knob == 0 && value >= 10) || ((knob == 4 || knob == 6 || knob == 17) && value != 0);
    if (wrong) {
        const char* ok = getenv("VATL_ALLOW_ABLATION");
        if (!ok || ok[0] != '1') return fail(VATL_EINVAL, "tune_set: knob %d value %d is a profiling ablation (wrong results); set VATL_ALLOW_ABLATION=1", knob, value);
    }
    if (knob == 0 && value >= 10 && value <= 13) { g_var.store(value, std::memory_order_relaxed); return 0; }
    if (knob == 2 && value >= 0 && value <= 200) { g_stagger.store(value, std::memory_order_relaxed); return 0; }
    if (knob == 3 && tune_wgrad_blocks(value) == 0) return 0;
    if (knob == 4 && value >= 0 && value <= 3) return tune_wgrad_blocks(-value - 1);
    if (knob == 6 && value >= 0 && value <= 15) { g_ablate.store(value, std::memory_order_relaxed); return 0; }
    if (knob == 9 && (value == 0 || value == 1)) { g_splitk_policy.store(value, std::memory_order_relaxed); return 0; }
    if (knob == 12 && (value == 0 || value == 1)) { g_streamk.store(value, std::memory_order_relaxed); return 0; }
    if (knob == 16 && crop_tune_px(value) == 0) return 0;
    if (knob == 17 && value >= 0 && value <= 63) return wino_set_ablate(value);
    if (knob == 19 && value >= 1 && value <= (1 << 20)) return wino_wgrad_set_blocks(value);
    if (knob == 23 && value >= 1 && value <= 2) return wino_wgrad_set_halves(value);
    return fail(VATL_EINVAL, "tune_set: unknown knob %d / value %d", knob, value);
#else
    return fail(VATL_EINVAL, "tune_set: knob %d / value %d is not a product knob (include/vatl_hip.h lists them: 0, 1, 5, 7, 8, 10, 18, 21, 22, 24, 25 — all bit-identical); "
                "knobs that change the summation order, performance experiments and profiling ablations exist only in the variant built with "
                "`build.py --ablation` (-DVATL_ABLATION)", knob, value);
#endif
}

extern "C" int vatl_set_splitk_workspace_thread(float* workspace, int64_t floats) {
    if (workspace && floats <= 0) return fail(VATL_EINVAL, "set_splitk_workspace_thread: empty workspace");
    tl_splitk_ws = workspace;
    tl_splitk_floats = workspace ? (long long)floats : 0;
    return 0;
}

extern "C" int64_t vatl_streamk_workspace_bytes(void) { return kStreamKBytes; }

extern "C" int vatl_set_streamk_workspace_thread(void* workspace, int64_t bytes) {
    if (workspace && bytes < kStreamKBytes) return fail(VATL_EINVAL, "set_streamk_workspace_thread: %lld bytes, need %lld", (long long)bytes, kStreamKBytes);
    tl_streamk_ws = static_cast<char*>(workspace);
    return 0;
}

extern "C" int vatl_set_splitk_workspace(float* workspace, int64_t floats) {
    if (workspace && floats <= 0) return fail(VATL_EINVAL, "set_splitk_workspace: empty workspace");
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return fail(VATL_EINVAL, "set_splitk_workspace: no current device (or index >= %d)", kMaxDevices);
    g_splitk_floats[d].store(workspace ? (long long)floats : 0, std::memory_order_relaxed);
    g_splitk_ws[d].store(workspace, std::memory_order_release);
    int any = 0;
    for (int i = 0; i < kMaxDevices; ++i) any |= g_splitk_ws[i].load(std::memory_order_relaxed) != nullptr;
    g_splitk_any.store(any, std::memory_order_release);
    return 0;
}

extern "C" int vatl_conv_cout_pad(int Cout) {
    const int bn = tile_n_for(Cout);
    return (Cout + bn - 1) / bn * bn;
}

static int conv2d_fwd_impl(const float* x, const float* w, const float* scale, const float* bias, const float* residual,
                           float* y, int N, int H, int W, int Cin, int Cout, int CoutPad, int R, int S, int stride, int pad,
                           int relu, int out_nchw, double* stats, int64_t* row_blocks, void* stream) {
    if (!x || !w || !y || N <= 0) return fail(VATL_EINVAL, "conv2d_fwd: null pointer or empty batch");
    if (stats && ((Cout & 3) || out_nchw)) return fail(VATL_EINVAL, "conv2d_fwd_stats: Cout %d must be a multiple of 4 (NHWC output)", Cout);
    const bool stem = (Cin == 4);
    if (!stem && Cin % 32 != 0) return fail(VATL_EINVAL, "conv2d_fwd: Cin %d must be a multiple of 32 (or 4 for the stem)", Cin);
    if (!stats && !out_nchw) {                            // narrow 3x3 layers: the persistent halo-tile kernel (bit-identical results)
        const int rc = conv3x3_halo_try(x, w, scale, bias, residual, y, N, H, W, Cin, Cout, CoutPad, R, S, stride, pad, relu, (hipStream_t)stream);
        if (rc <= 0) return rc;                           // 1 = not one of its shapes: fall through to the implicit GEMM
    }
    if (stem && S > 8) return fail(VATL_EINVAL, "conv2d_fwd: stem filter width %d > 8", S);
    ConvParams p{};
    p.x = x; p.w = w; p.scale = scale; p.bias = bias; p.res = residual; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.CoutPad = CoutPad;
    p.R = R; p.S = S; p.stride = stride; p.pad_y = pad; p.pad_x = pad;
    p.Ho = (H + 2 * pad - R) / stride + 1;
    p.Wo = (W + 2 * pad - S) / stride + 1;
    p.M = N * p.Ho * p.Wo;
    p.OH = p.Ho; p.OW = p.Wo; p.osy = 1; p.osx = 1; p.ooy = 0; p.oox = 0;
    p.relu = relu; p.out_nchw = out_nchw; p.deconv = 0;
    if (stem) { p.kpr = 1; p.ktiles = R; p.K = R * 8 * 4; }
    else      { p.kpr = Cin / BK; p.ktiles = R * S * p.kpr; p.K = R * S * Cin; }
    const long long xe = (long long)N * H * W * Cin, ye = (long long)p.M * Cout, we = (long long)CoutPad * p.K;
    if (xe >= (1LL << 30) || ye >= (1LL << 30) || we >= (1LL << 30))
        return fail(VATL_EINVAL, "conv2d_fwd: a tensor exceeds 2^30 elements (32-bit buffer offsets); split the batch");
    p.x_bytes = (unsigned)(xe * 4); p.y_bytes = (unsigned)(ye * 4); p.w_bytes = (unsigned)(we * 4);
    p.stats = stats;
    return dispatch(p, 1, stem, (hipStream_t)stream, row_blocks);
}

extern "C" int vatl_conv2d_fwd(const float* x, const float* w, const float* scale, const float* bias, const float* residual,
                               float* y, int N, int H, int W, int Cin, int Cout, int CoutPad, int R, int S, int stride, int pad,
                               int relu, int out_nchw, void* stream) {
    return conv2d_fwd_impl(x, w, scale, bias, residual, y, N, H, W, Cin, Cout, CoutPad, R, S, stride, pad, relu, out_nchw, nullptr, nullptr, stream);
}

extern "C" int64_t vatl_conv_stats_row_blocks(int64_t gemm_rows, int phases) { return (gemm_rows + 63) / 64 * phases; }   // capacity

extern "C" int vatl_conv2d_fwd_stats(const float* x, const float* w, float* y, double* stats, int64_t* row_blocks_used, int N, int H, int W,
                                     int Cin, int Cout, int CoutPad, int R, int S, int stride, int pad, void* stream) {
    if (!stats || !row_blocks_used) return fail(VATL_EINVAL, "conv2d_fwd_stats: null statistics buffer");
    return conv2d_fwd_impl(x, w, nullptr, nullptr, nullptr, y, N, H, W, Cin, Cout, CoutPad, R, S, stride, pad, 0, 0, stats, row_blocks_used, stream);
}

static int deconv4x4s2_fwd_impl(const float* x, const float* w, const float* scale, const float* bias, float* y,
                                int N, int H, int W, int Cin, int Cout, int CoutPad, int relu, double* stats, int64_t* row_blocks, void* stream) {
    if (!x || !w || !y || N <= 0) return fail(VATL_EINVAL, "deconv4x4s2_fwd: null pointer or empty batch");
    if (stats && (Cout & 3)) return fail(VATL_EINVAL, "deconv4x4s2_fwd_stats: Cout %d must be a multiple of 4", Cout);
    if (Cin % 32 != 0) return fail(VATL_EINVAL, "deconv4x4s2_fwd: Cin %d must be a multiple of 32", Cin);
    ConvParams p{};
    p.x = x; p.w = w; p.scale = scale; p.bias = bias; p.res = nullptr; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.CoutPad = CoutPad;
    p.R = 2; p.S = 2; p.stride = 1; p.pad_y = 0; p.pad_x = 0;
    p.Ho = H; p.Wo = W; p.M = N * H * W;
    p.OH = 2 * H; p.OW = 2 * W; p.osy = 2; p.osx = 2; p.ooy = 0; p.oox = 0;
    p.relu = relu; p.out_nchw = 0; p.deconv = 1;
    p.kpr = Cin / BK; p.ktiles = 4 * p.kpr; p.K = 4 * Cin;
    const long long xe = (long long)N * H * W * Cin, ye = 4LL * p.M * Cout, we = (long long)CoutPad * p.K;
    if (xe >= (1LL << 30) || ye >= (1LL << 30) || 4 * we >= (1LL << 30))
        return fail(VATL_EINVAL, "deconv4x4s2_fwd: a tensor exceeds 2^30 elements (32-bit buffer offsets); split the batch");
    p.x_bytes = (unsigned)(xe * 4); p.y_bytes = (unsigned)(ye * 4); p.w_bytes = (unsigned)(we * 4);   // w: one phase
    p.stats = stats;
    return dispatch(p, 4, false, (hipStream_t)stream, row_blocks);
}

extern "C" int vatl_deconv4x4s2_fwd(const float* x, const float* w, const float* scale, const float* bias, float* y,
                                    int N, int H, int W, int Cin, int Cout, int CoutPad, int relu, void* stream) {
    return deconv4x4s2_fwd_impl(x, w, scale, bias, y, N, H, W, Cin, Cout, CoutPad, relu, nullptr, nullptr, stream);
}

extern "C" int vatl_deconv4x4s2_fwd_stats(const float* x, const float* w, float* y, double* stats, int64_t* row_blocks_used, int N, int H, int W,
                                          int Cin, int Cout, int CoutPad, void* stream) {
    if (!stats || !row_blocks_used) return fail(VATL_EINVAL, "deconv4x4s2_fwd_stats: null statistics buffer");
    return deconv4x4s2_fwd_impl(x, w, nullptr, nullptr, y, N, H, W, Cin, Cout, CoutPad, 0, stats, row_blocks_used, stream);
}

// Last 1x1 conv of a bottleneck fused with the block's projection shortcut (Resnet.py:104-128 with a `downsample`):
//   y = act( W1' a  +  W2' x[strided]  +  bias ),   a (N,Ho,Wo,C1) = output of the 3x3, x (N,H2,W2,C2) = block input,
// one implicit GEMM over K = C1 + C2 whose A operand comes from two tensors.  w: [CoutPad][C1 + C2] with the two folded
// BatchNorm scales already multiplied into the rows (vatl_pack_conv1x1_dual_weight); bias = bias1 + bias2.  The
// projection's output never exists in HBM (for layer 1 of a ResNet-50 that is 3.2 GB written + read per 1024 crops).
extern "C" int vatl_conv1x1_dual_fwd(const float* a, const float* x, const float* w, const float* bias, float* y, int N, int Ho, int Wo,
                                     int C1, int H2, int W2, int C2, int stride2, int Cout, int CoutPad, int relu, void* stream) {
    if (!a || !x || !w || !y || N <= 0) return fail(VATL_EINVAL, "conv1x1_dual_fwd: null pointer or empty batch");
    if ((C1 % 32) || (C2 % 32)) return fail(VATL_EINVAL, "conv1x1_dual_fwd: channel counts %d / %d must be multiples of 32", C1, C2);
    if (Cout < 128 || (Cout & 3)) return fail(VATL_EINVAL, "conv1x1_dual_fwd: Cout %d must be >= 128 and a multiple of 4", Cout);
    if (stride2 < 1 || (Ho - 1) * stride2 >= H2 || (Wo - 1) * stride2 >= W2) return fail(VATL_EINVAL, "conv1x1_dual_fwd: stride %d does not map %dx%d onto %dx%d", stride2, Ho, Wo, H2, W2);
    ConvParams p{};
    p.x = a; p.x2 = x; p.w = w; p.scale = nullptr; p.bias = bias; p.res = nullptr; p.y = y;
    p.N = N; p.H = Ho; p.W = Wo; p.Cin = C1; p.Cout = Cout; p.CoutPad = CoutPad;
    p.R = 1; p.S = 1; p.stride = 1; p.pad_y = 0; p.pad_x = 0;
    p.Ho = Ho; p.Wo = Wo; p.M = N * Ho * Wo;
    p.OH = Ho; p.OW = Wo; p.osy = 1; p.osx = 1; p.ooy = 0; p.oox = 0;
    p.relu = relu; p.out_nchw = 0; p.deconv = 0;
    p.k1 = C1 / BK; p.C2 = C2; p.H2 = H2; p.W2 = W2; p.stride2 = stride2;
    p.kpr = (C1 + C2) / BK; p.ktiles = p.kpr; p.K = C1 + C2;
    const long long xe = (long long)p.M * C1, x2e = (long long)N * H2 * W2 * C2, ye = (long long)p.M * Cout, we = (long long)CoutPad * p.K;
    if (xe >= (1LL << 30) || x2e >= (1LL << 30) || ye >= (1LL << 30) || we >= (1LL << 30))
        return fail(VATL_EINVAL, "conv1x1_dual_fwd: a tensor exceeds 2^30 elements (32-bit buffer offsets); split the batch");
    p.x_bytes = (unsigned)(xe * 4); p.x2_bytes = (unsigned)(x2e * 4); p.y_bytes = (unsigned)(ye * 4); p.w_bytes = (unsigned)(we * 4);
    if (CoutPad % 128) return fail(VATL_EINVAL, "conv1x1_dual_fwd: CoutPad %d must be a multiple of 128", CoutPad);
    const int bm = tile_m_for(p, 1, 128, false);
    p.d_HoWo = make_fastdiv((unsigned)(p.Ho * p.Wo)); p.d_Wo = make_fastdiv((unsigned)p.Wo);
    if (bm == 64) return launch<64, 128, 32, 64, false, 4, true>(p, 1, (hipStream_t)stream);
    return launch<128, 128, 64, 64, false, 4, true>(p, 1, (hipStream_t)stream);
}

// General form behind the data-gradient paths: explicit GEMM pixel grid (Ho x Wo), separate paddings and an
// output scatter (oy*osy+ooy, ox*osx+oox) into an OH x OW image.  Transposed-conv style gradients are
// phase-decomposed by the caller (alphapose/models/hip_train.py): e.g. the data gradient of a 3x3/2 conv is
// four launches with 1x1 / 1x2 / 2x1 / 2x2 taps over the output-gradient grid, scattered with osy = osx = 2.
static int conv2d_fwd_ex_impl(const float* x, const float* w, const float* scale, const float* bias, const float* residual, float* y,
                              int N, int H, int W, int Cin, int Cout, int CoutPad, int R, int S, int stride, int pad_y, int pad_x,
                              int Ho, int Wo, int OH, int OW, int osy, int osx, int ooy, int oox, int relu, const float* bn_z, const float* bn_mask_y,
                              const float* bn_scale, const float* bn_bias, const float* bn_mean, const float* bn_invstd, double* stats,
                              int64_t* row_blocks_used, void* stream) {
    if (!x || !w || !y || N <= 0) return fail(VATL_EINVAL, "conv2d_fwd_ex: null pointer or empty batch");
    if (Cin % 32 != 0) return fail(VATL_EINVAL, "conv2d_fwd_ex: Cin %d must be a multiple of 32", Cin);
    ConvParams p{};
    p.x = x; p.w = w; p.scale = scale; p.bias = bias; p.res = residual; p.y = y;
    p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.CoutPad = CoutPad;
    p.R = R; p.S = S; p.stride = stride; p.pad_y = pad_y; p.pad_x = pad_x;
    p.Ho = Ho; p.Wo = Wo; p.M = N * Ho * Wo;
    p.OH = OH; p.OW = OW; p.osy = osy; p.osx = osx; p.ooy = ooy; p.oox = oox;
    p.relu = relu; p.out_nchw = 0; p.deconv = 0;
    p.kpr = Cin / BK; p.ktiles = R * S * p.kpr; p.K = R * S * Cin;
    const long long xe = (long long)N * H * W * Cin, ye = (long long)N * OH * OW * Cout, we = (long long)CoutPad * p.K;
    if (xe >= (1LL << 30) || ye >= (1LL << 30) || we >= (1LL << 30))
        return fail(VATL_EINVAL, "conv2d_fwd_ex: a tensor exceeds 2^30 elements (32-bit buffer offsets); split the batch");
    p.x_bytes = (unsigned)(xe * 4); p.y_bytes = (unsigned)(ye * 4); p.w_bytes = (unsigned)(we * 4);
    if (bn_z) {
        if (!stats || !row_blocks_used || !bn_mean || !bn_invstd || relu || scale || bias || (bn_scale && !bn_bias))
            return fail(VATL_EINVAL, "conv2d_fwd_ex_bnbwd: needs z, mean, invstd, a statistics buffer; no scale / bias / ReLU of its own");
        p.bz = bn_z; p.bmy = bn_mask_y; p.bsc = bn_scale; p.bbi = bn_bias; p.bmu = bn_mean; p.bis = bn_invstd; p.stats = stats;
    }
    return dispatch(p, 1, false, (hipStream_t)stream, row_blocks_used);
}

extern "C" int vatl_conv2d_fwd_ex(const float* x, const float* w, const float* scale, const float* bias, const float* residual, float* y,
                                  int N, int H, int W, int Cin, int Cout, int CoutPad, int R, int S, int stride, int pad_y, int pad_x,
                                  int Ho, int Wo, int OH, int OW, int osy, int osx, int ooy, int oox, int relu, void* stream) {
    return conv2d_fwd_ex_impl(x, w, scale, bias, residual, y, N, H, W, Cin, Cout, CoutPad, R, S, stride, pad_y, pad_x, Ho, Wo, OH, OW, osy, osx, ooy, oox,
                              relu, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, stream);
}

// Data-gradient launch fused with the reduction pass of a BatchNorm backward (ActiveLearning.py:672, loss.backward()): y receives
// g = (W^T dz + residual) * [consumer layer's ReLU mask] and `stats` the per-(row block, channel) double partials of
// (sum g, sum g * xhat) in the layout vatl_bn_bwd_from_stats reduces.  bn_z = that layer's conv output (layout of y);
// mask = bn_mask_y > 0 when given (layers whose ReLU follows a residual sum), else bn_z*bn_scale+bn_bias > 0 when bn_scale is
// given, else none.  row_blocks_used returns how many row blocks were written (<= vatl_conv_stats_row_blocks(N*Ho*Wo, 1)).
extern "C" int vatl_conv2d_fwd_ex_bnbwd(const float* x, const float* w, const float* residual, float* y, int N, int H, int W, int Cin, int Cout,
                                        int CoutPad, int R, int S, int stride, int pad_y, int pad_x, int Ho, int Wo, int OH, int OW, int osy, int osx,
                                        int ooy, int oox, const float* bn_z, const float* bn_mask_y, const float* bn_scale, const float* bn_bias,
                                        const float* bn_mean, const float* bn_invstd, double* stats, int64_t* row_blocks_used, void* stream) {
    if (!bn_z) return fail(VATL_EINVAL, "conv2d_fwd_ex_bnbwd: null bn_z");
    return conv2d_fwd_ex_impl(x, w, nullptr, nullptr, residual, y, N, H, W, Cin, Cout, CoutPad, R, S, stride, pad_y, pad_x, Ho, Wo, OH, OW, osy, osx, ooy,
                              oox, 0, bn_z, bn_mask_y, bn_scale, bn_bias, bn_mean, bn_invstd, stats, row_blocks_used, stream);
}
