// Winograd F(2x2, 3x3) for the 64 -> 64 channel 3x3 / stride 1 / pad 1 layers (HRNet's second branch: 64 BasicBlock convs per pass, hrnet.py:24-56; conv2 of
// ResNet's stage 1, Resnet.py:104-128) with wave-private tiles, like csrc/winograd_c32.hip, at ONE wave per SIMD:
//   * a wave owns a 4 x 4 patch of tiles with all 16 transform positions and all 64 output channels: 16 x 4 accumulator tiles of v_mfma_f32_16x16x4_f32 = 256
//     registers, which only a wave that has the SIMD's register file to itself can hold (512 per lane: the accumulators live in the AGPR half).  One input
//     transform serves four 16-column blocks and one output transform 16 K-steps: half the vector instructions per MFMA of the 32-channel kernel — and vector
//     instructions are paid in full next to fp32 MFMAs on this hardware (tools/probes/mfma_valu_overlap.hip), while a second wave per SIMD buys little
//     (winograd_c32 with one wave per SIMD: 318 - 332 us against 286);
//   * the transformed filter (256 KB) does not fit the LDS, and a shared LDS ring costs a barrier per K-step (measured: 10 % of the kernel): every wave loads
//     its filter fragments itself, global -> registers, one K-step ahead (16 KB per K-step; the four waves of a CU ask for the same lines within a K-step, the
//     filter stays in L2) — no barrier anywhere after the prologue;
//   * everything else is private to the wave too: its 10 x 10-pixel input patch (LDS-DMA, 16 channels at a time, two buffers), the input transform B^T d B in
//     registers (lane = (tile, channel of the K-step's four)), the output transform A^T M A and the write-out from the accumulators.
// The MFMAs take the FILTER as their A operand: a lane's accumulator tuple is then four consecutive output channels of one pixel, and the write-out (and the
// skip-connection read) is 16 wide memory operations per unit instead of 64 narrow ones.
#include "common.h"

#include <atomic>
#include <utility>

namespace vatl {

struct C64Params {
    const float* x;          // (N, H, W, 64)
    const float* u;          // packed filter [16 K-steps][4 column blocks][4 position rows xi][64 lanes][4 nu]
    const float* scale;      // (64) or null
    const float* bias;       // (64) or null
    const float* res;        // (N, H, W, 64) or null
    float* y;                // (N, H, W, 64)
    int N, H, W, TH, TW, UH, UW, relu;
    int units, upi, iters;   // units (4 x 4 tile patches) in the launch / per image; unit rounds per block
    unsigned bytes;          // of x / y / res
    FastDivU d_upi, d_uw;
};

typedef __attribute__((address_space(3))) void c64_lds_void;
typedef unsigned c64_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned C64_OOB = 0xFFFFFFF0u;
constexpr int C64_U_FLOATS = 16 * 4 * 4 * 64 * 4;   // 65536
constexpr int C64_PATCH = 7 * 256;                  // floats of a patch buffer: 10 x 10 pixels x 16 channels = 400 16-byte chunks, 7 requests of 64
constexpr int C64_LDS_FLOATS = 4 * 2 * C64_PATCH + 128;

template <int N> __device__ __forceinline__ void c64_wait() {
    static_assert(N >= 0 && N <= 63, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// K-steps with a compile-time index (the counted waits are immediates)
template <class F, int... I> __device__ __forceinline__ void c64_unroll(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }

// (body in a __device__ function: with the DMA builtin inside a __global__ template hipcc 7.2 drops the kernel's host stub)
template <bool RES>
__device__ __forceinline__ void winograd_c64_body(const C64Params& p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Pw = smem + wave * 2 * C64_PATCH;
    float* SBs = smem + 4 * 2 * C64_PATCH;         // scale[64], bias[64]
    if (tid < 64) { SBs[tid] = p.scale ? p.scale[tid] : 1.f; SBs[64 + tid] = p.bias ? p.bias[tid] : 0.f; }

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.res : p.y), 0, RES ? p.bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t ur = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, C64_U_FLOATS * 4, 0x00020000);

    // ---- staging constants, in REGISTERS (an LDS table read behind an LDS-DMA request makes the compiler wait for every request in flight: it cannot tell the table
    // from the DMA's destination): request j writes LDS chunks 64 j .. 64 j + 63; chunk id = pixel * 4 + position, position = source quad ^ swizzle, swizzle =
    // (patch row >> 1) & 3 (the 16 tiles of a read fall on 8 different 4-word bank groups).  rel[j]: byte offset of the chunk's source from the patch origin;
    // bit j of ok_all / ok_top / ok_bottom / ok_left / ok_right: the chunk exists / is inside the image when the unit touches that border ----
    int rel[7];
    int ok_all = 0, ok_top = 0, ok_bottom = 0, ok_left = 0, ok_right = 0;
    {
        const int ylast = p.H - (8 * (p.UH - 1) - 1), xlast = p.W - (8 * (p.UW - 1) - 1);              // patch rows / columns below these are inside the image
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int id = 64 * j + lane, px = id >> 2;
            const int pr = px / 10, pc = px - pr * 10;
            const int sq = (id & 3) ^ ((pr >> 1) & 3);
            rel[j] = ((pr * p.W + pc) * 64 + sq * 4) * 4;
            ok_all |= (px < 100 ? 1 : 0) << j;
            ok_top |= (pr >= 1 ? 1 : 0) << j;
            ok_bottom |= (pr < ylast ? 1 : 0) << j;
            ok_left |= (pc >= 1 ? 1 : 0) << j;
            ok_right |= (pc < xlast ? 1 : 0) << j;
        }
    }
    __syncthreads();

    // input transform: lane (tile t = lane % 16 at (t & 3, t >> 2) of the 4 x 4 patch of tiles, channel k4 = lane / 16 of the K-step's four).  LDS word of pixel
    // (2 tyl + i, 2 txl + j), K-step s of the quarter: (pixel * 4 + (s ^ swizzle of the pixel's row)) * 4 + k4: two swizzles per lane (rows i < 2 / i >= 2)
    const int t16 = lane & 15, k4 = lane >> 4;
    const int txl = t16 & 3, tyl = t16 >> 2;
    const int d_base = (2 * tyl * 10 + 2 * txl) * 16 + k4;
    const int swz_a = tyl & 3, swz_b = (tyl + 1) & 3;
    // write-out: the MFMAs run with the FILTER as the A operand, so a lane's accumulator tuple is (tile lane % 16) x (couts 16 nb + 4 (lane / 16) + 0 .. 3): 16 bytes
    // of one NHWC pixel — 16 wide stores (and skip-connection loads) per unit instead of 64 narrow ones (a wave has at most 63 memory operations in flight)
    const float lo = p.relu ? 0.f : -INFINITY;
    const unsigned lane16 = (unsigned)lane * 16u;

    auto stage = [&](int unit, int quarter) {              // 16 channels of the unit's patch -> this wave's buffer quarter & 1 (zeros outside the image = the padding)
        const bool live = unit < p.units;
        const int img = fdiv(unit, p.d_upi), rem = unit - img * p.upi;
        const int uy = fdiv(rem, p.d_uw), ux = rem - uy * p.UW;
        const int okbits = live ? (ok_all & (uy == 0 ? ok_top : -1) & (uy == p.UH - 1 ? ok_bottom : -1) & (ux == 0 ? ok_left : -1) & (ux == p.UW - 1 ? ok_right : -1)) : 0;
        const int origin = (((img * p.H + 8 * uy - 1) * p.W + 8 * ux - 1) * 64 + quarter * 16) * 4;
        float* buf = Pw + (quarter & 1) * C64_PATCH;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const unsigned off = (okbits >> j) & 1 ? (unsigned)(origin + rel[j]) : C64_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (c64_lds_void*)(buf + j * 256), 16, off, 0, 0, 0);
        }
    };

    f32x4 uf[4][4];                                        // filter fragments [column block][position row xi] of the K-step about to run
    auto load_uf = [&](int S, int nb) {
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) uf[nb][xi] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, lane16, ((S * 4 + nb) * 4 + xi) * 1024, 0));
    };
    float d[16];
    auto load_d = [&](int S) {                             // the 16 pixels of this lane's (tile, channel) for K-step S (of the unit whose quarter S / 4 is in buffer (S / 4) & 1)
        const float* buf = Pw + ((S >> 2) & 1) * C64_PATCH + d_base;
        const float* pa = buf + ((swz_a ^ (S & 3)) << 2);
        const float* pb = buf + ((swz_b ^ (S & 3)) << 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i * 4 + j] = (i < 2 ? pa : pb)[(i * 10 + j) * 16];
    };

    const int ustride = gridDim.x * 4;
    int unit = blockIdx.x * 4 + wave;
    stage(unit, 0);
    c64_wait<0>();
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) load_uf(0, nb);

    for (int it = 0; it < p.iters; ++it, unit += ustride) {
        f32x4 acc[16][4];
        f32x4 yres[RES ? 16 : 1];
        const int img = fdiv(unit, p.d_upi), rem = unit - img * p.upi;
        const int uy = fdiv(rem, p.d_uw), ux = rem - uy * p.UW;
        const int ty = 4 * uy + tyl, tx = 4 * ux + txl;
        // byte offset of this lane's tile's first output pixel, couts 4 k4 .. of column block 0
        const unsigned base = (unit < p.units && ty < p.TH && tx < p.TW) ? (unsigned)((((img * p.H + 2 * ty) * p.W + 2 * tx) * 64 + 4 * k4) * 4) : C64_OOB;
        const unsigned rowb = (unsigned)(p.W * 64 * 4);
        // this unit's first patch quarter (requested in K-step 12 of the unit before): at least 64 younger memory operations have been issued since
        c64_wait<63>();
        load_d(0);
        auto kstep = [&](auto Sc) {
            constexpr int S = decltype(Sc)::value;
            if ((S & 3) == 0) stage(S < 12 ? unit : unit + ustride, ((S >> 2) + 1) & 3);
            if (RES && S == 13) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                    for (int ab = 0; ab < 4; ++ab)
                        yres[nb * 4 + ab] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rr, base != C64_OOB ? base + (ab >> 1) * rowb + (ab & 1) * 256u + nb * 64u : C64_OOB, 0, 0));
            }
            // V = B^T d B on register pairs (as in winograd_c32)
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 dl[4], dh[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { dl[i] = f32x2{d[i * 4 + 0], d[i * 4 + 1]}; dh[i] = f32x2{d[i * 4 + 2], d[i * 4 + 3]}; }
            const f32x2 tl[4] = {dl[0] - dl[2], dl[1] + dl[2], dl[2] - dl[1], dl[1] - dl[3]};
            const f32x2 th[4] = {dh[0] - dh[2], dh[1] + dh[2], dh[2] - dh[1], dh[1] - dh[3]};
            float v[16];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                f32x2 v01, v23;
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(v01) : "v"(tl[i]), "v"(th[i]));
                asm("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[1,0] neg_hi:[0,1]" : "=v"(v23) : "v"(tl[i]), "v"(th[i]));
                v[i * 4 + 0] = v01.x; v[i * 4 + 1] = v01.y; v[i * 4 + 2] = v23.x; v[i * 4 + 3] = v23.y;
            }
            // the next K-step's pixels; a new quarter's patch was requested at the top of K-step S - 3, before the 48 fragment loads of K-steps S - 3 .. S - 1
            if (S < 15 && (S & 3) == 3) c64_wait<48>();
            if (S < 15) load_d(S + 1);
            __builtin_amdgcn_sched_barrier(0);
            // MFMAs as inline assembly with the accumulator tied to an AGPR tuple: with the builtin the register allocator treats an MFMA's input and output
            // accumulator as two live ranges, fragments the (exactly full) AGPR half and shuffles tuples through VGPRs between the MFMAs.  What the compiler
            // then no longer sees is the MFMA -> reader hazard: the same accumulator is only touched again 64 MFMAs later, and the epilogue starts with its own
            // wait states.
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (S == 0) asm("v_mfma_f32_16x16x4_f32 %0, %2, %1, 0" : "=a"(acc[q][nb]) : "v"(v[q]), "v"(uf[nb][q >> 2][q & 3]));
                    else asm("v_mfma_f32_16x16x4_f32 %0, %2, %1, %0" : "+a"(acc[q][nb]) : "v"(v[q]), "v"(uf[nb][q >> 2][q & 3]));
                }
                if (S < 15) load_uf(S + 1, nb);            // the next K-step's fragments of this column block: a K-step of MFMAs ahead
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        c64_unroll(kstep, std::make_integer_sequence<int, 16>{});
        // ---- output transform (registers only) + write-out: tile lane % 16, couts 16 nb + 4 k4 + 0 .. 3 ----
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (8 passes) before the first accumulator read
        __builtin_amdgcn_sched_barrier(0);
        if (!RES) {                                        // the next unit's first fragments, behind the write-out (with a skip connection: once half its values are used)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) load_uf(0, nb);
        }
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            if (RES && nb == 2) {
#pragma unroll
                for (int n2 = 0; n2 < 4; ++n2) load_uf(0, n2);
            }
            const f32x4 scn = *reinterpret_cast<const f32x4*>(SBs + 16 * nb + 4 * k4), bin = *reinterpret_cast<const f32x4*>(SBs + 64 + 16 * nb + 4 * k4);
            f32x4 t0[4], t1[4], yq[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                t0[nu] = acc[0 * 4 + nu][nb] + acc[1 * 4 + nu][nb] + acc[2 * 4 + nu][nb];
                t1[nu] = acc[1 * 4 + nu][nb] - acc[2 * 4 + nu][nb] - acc[3 * 4 + nu][nb];
            }
            yq[0] = (t0[0] + t0[1] + t0[2]) * scn + bin;
            yq[1] = (t0[1] - t0[2] - t0[3]) * scn + bin;
            yq[2] = (t1[0] + t1[1] + t1[2]) * scn + bin;
            yq[3] = (t1[1] - t1[2] - t1[3]) * scn + bin;
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                f32x4 o = yq[ab];
                if (RES) o += yres[nb * 4 + ab];
                o = f32x4{fmaxf(o[0], lo), fmaxf(o[1], lo), fmaxf(o[2], lo), fmaxf(o[3], lo)};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(c64_u32x4, o), yr, base != C64_OOB ? base + (ab >> 1) * rowb + (ab & 1) * 256u + nb * 64u : C64_OOB, 0, 0);
            }
        }
    }
    c64_wait<0>();                                         // no request of this block may land in the LDS of the next
}

template <bool RES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void winograd_c64_kernel(C64Params p) {
    winograd_c64_body<RES>(p);
}

// (Cout = 64, Cin = 64, 3, 3) -> U = G g G^T in the kernel's ring order: [K-step S][column block nb][xi][lane = n + 16 k][nu] (position p = 4 xi + nu), channel 4 S + k, cout 16 nb + n
__global__ __launch_bounds__(256) void winograd_c64_pack_kernel(const float* __restrict__ w, float* __restrict__ u) {
    const int id = blockIdx.x * 256 + threadIdx.x;        // (S, nb, lane)
    if (id >= 16 * 4 * 64) return;
    const int lane = id & 63, nb = (id >> 6) & 3, S = id >> 8;
    const int c = 4 * S + (lane >> 4), n = 16 * nb + (lane & 15);
    const float* g = w + ((long long)n * 64 + c) * 9;
    float gg[4][3];                                       // G g: rows (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gg[0][j] = g[0 * 3 + j];
        gg[1][j] = 0.5f * (g[0 * 3 + j] + g[1 * 3 + j] + g[2 * 3 + j]);
        gg[2][j] = 0.5f * (g[0 * 3 + j] - g[1 * 3 + j] + g[2 * 3 + j]);
        gg[3][j] = g[2 * 3 + j];
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        float* o = u + ((((S * 4 + nb) * 4 + xi) * 64 + lane) * 4);
        o[0] = gg[xi][0];
        o[1] = 0.5f * (gg[xi][0] + gg[xi][1] + gg[xi][2]);
        o[2] = 0.5f * (gg[xi][0] - gg[xi][1] + gg[xi][2]);
        o[3] = gg[xi][2];
    }
}

}  // namespace vatl

using namespace vatl;

extern "C" int64_t vatl_winograd_c64_weight_floats(void) { return C64_U_FLOATS; }

extern "C" int vatl_pack_winograd_c64_weight(const float* w, float* u, void* stream) {
    if (!w || !u) return fail(VATL_EINVAL, "pack_winograd_c64_weight: null pointer");
    hipLaunchKernelGGL(winograd_c64_pack_kernel, dim3(16), dim3(256), 0, (hipStream_t)stream, w, u);
    return check_launch("winograd_c64_pack");
}

extern "C" int vatl_conv3x3_winograd_c64_supported(int N, int H, int W, int Cin, int Cout) {
    return Cin == 64 && Cout == 64 && N > 0 && H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0 && (long long)N * H * W * 64 < (1LL << 29) ? 1 : 0;
}

extern "C" int vatl_conv3x3_winograd_c64_fwd(const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y, int N, int H, int W,
                                             int relu, void* stream) {
    if (!x || !u || !y) return fail(VATL_EINVAL, "conv3x3_winograd_c64_fwd: null pointer");
    if (!vatl_conv3x3_winograd_c64_supported(N, H, W, 64, 64)) return fail(VATL_EINVAL, "conv3x3_winograd_c64_fwd: serves 64 -> 64 channels, even H and W, N * H * W * 64 < 2^29");
    C64Params p{};
    p.x = x; p.u = u; p.scale = scale; p.bias = bias; p.res = residual; p.y = y;
    p.N = N; p.H = H; p.W = W; p.TH = H / 2; p.TW = W / 2; p.UH = (p.TH + 3) / 4; p.UW = (p.TW + 3) / 4; p.relu = relu;
    p.upi = p.UH * p.UW; p.units = N * p.upi;
    p.bytes = (unsigned)((long long)N * H * W * 64 * 4);
    p.d_upi = make_fastdiv((unsigned)p.upi); p.d_uw = make_fastdiv((unsigned)p.UW);
    int grid = (p.units + 3) / 4;
    if (grid > 256) grid = 256;                            // one block per CU
    p.iters = (p.units + 4 * grid - 1) / (4 * grid);
    const int smem = C64_LDS_FLOATS * (int)sizeof(float);
    static std::atomic<unsigned> configured[2] = {{0}, {0}};
    if (residual) {
        auto kern = winograd_c64_kernel<true>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, configured[1], "winograd_c64")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, (hipStream_t)stream, p);
    } else {
        auto kern = winograd_c64_kernel<false>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, configured[0], "winograd_c64")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, (hipStream_t)stream, p);
    }
    // executed MFMA FLOPs: units x 16 tiles x 64 cout x 64 cin x 16 positions
    meter_add(1, 2.0 * (double)p.units * 16.0 * 64.0 * 64.0 * 16.0);
    return check_launch("winograd_c64");
}
