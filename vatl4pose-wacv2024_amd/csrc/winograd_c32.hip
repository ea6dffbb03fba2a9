// Winograd F(2x2, 3x3) for the 32 -> 32 channel 3x3 / stride 1 / pad 1 layers (HRNet's highest-resolution branch: 64 BasicBlock convs per pass, hrnet.py:24-56)
// with NO cross-wave exchange.  The general kernel (csrc/conv_winograd.hip) gives each of a block's four waves four of the 16 transform positions of 32 tiles, so
// the output transform needs an LDS round trip and two barriers per block, and a 32-channel block has 64 MFMAs to hide ~550 vector instructions behind
// (profiles/r04_notes.md: matrix pipe 45 % busy).  Here a WAVE owns 16 tiles (an 8 x 2 patch) completely, on v_mfma_f32_16x16x4_f32:
//   * A operand of position p = V_p[tile][channel]: lane (tile = lane % 16, k = lane / 16) computes B^T d B of ITS (tile, channel 4 s + k) from 16 LDS words
//     and holds all 16 positions — exactly the 16 A operands of K-step s, no shuffles;
//   * B operand U_p[channel][cout] comes from LDS, where the whole transformed filter (16 x 32 x 32 floats = 64 KB) lives for the block's life;
//   * the accumulators of a lane are (4 tiles) x (cout = lane % 16 of each 16-column block) x all 16 positions: the output transform A^T M A is register
//     arithmetic, the write-out goes straight to HBM (64-byte runs per pixel and column block).
// One block of eight waves per CU (two per SIMD); a wave stages its own 6 x 18-pixel input patch by LDS-DMA (16 channels at a time: 7 KB) and never meets a barrier
// after the filter is in LDS.  Per K-step a wave issues 16 + 8 LDS reads and ~50 vector instructions around 32 MFMAs of 32 cycles.
#include "common.h"

#include <atomic>

namespace vatl {

struct C32Params {
    const float* x;          // (N, H, W, 32)
    const float* u;          // packed filter [8 K-steps][2 column blocks][4 position rows xi][64 lanes][4 nu]
    const float* scale;      // (32) or null
    const float* bias;       // (32) or null
    const float* res;        // (N, H, W, 32) or null
    float* y;                // (N, H, W, 32)
    int N, H, W, TH, TW, UH, UW, relu;
    int units, upi;          // units (8 x 2 tile patches) in the launch / per image
    unsigned bytes;          // of x / y / res
    FastDivU d_upi, d_uw;
};

typedef __attribute__((address_space(3))) void c32_lds_void;
constexpr unsigned C32_OOB = 0xFFFFFFF0u;
constexpr int C32_U_FLOATS = 8 * 2 * 64 * 16;     // 16384
constexpr int C32_PATCH = 7 * 256;                // floats of a wave's patch buffer: 6 x 18 pixels x 16 channels = 432 16-byte chunks, 7 requests of 64
constexpr int C32_WAVES = 8;
constexpr int C32_LDS_FLOATS = C32_U_FLOATS + C32_WAVES * C32_PATCH + 64;       // + scale, bias

__global__ __launch_bounds__(512, 1) void winograd_c32_kernel(C32Params p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Us = smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Ps = smem + C32_U_FLOATS + wave * C32_PATCH;
    {   // the transformed filter: 64 KB, once
        const f32x4* src = reinterpret_cast<const f32x4*>(p.u);
        f32x4* dst = reinterpret_cast<f32x4*>(Us);
#pragma unroll
        for (int k = 0; k < C32_U_FLOATS / 4 / 512; ++k) dst[tid + 512 * k] = src[tid + 512 * k];
    }
    float* SBs = smem + C32_U_FLOATS + C32_WAVES * C32_PATCH;     // scale[32], bias[32]: read per unit (four registers less across the K-steps)
    if (tid < 32) { SBs[tid] = p.scale ? p.scale[tid] : 1.f; SBs[32 + tid] = p.bias ? p.bias[tid] : 0.f; }
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0, p.res ? p.bytes : 0u, 0x00020000);

    // ---- per-lane constants ----
    // staging: request j writes LDS chunks 64 j .. 64 j + 63; chunk id = pixel * 4 + position, position = source quad ^ ((patch column >> 2) & 3): the 16 tiles
    // of a read fall on 8 different 4-word bank groups (two-way conflicts; a padded pixel pitch of five chunks makes them 16 — measured: no faster, two more
    // requests per stage).  What a request needs per unit is origin + a per-lane constant and a yes / no for "inside the image", which depends only on whether the
    // unit touches the top / bottom / left / right border: both are per-lane constants (below), so a stage costs ~4 vector instructions per request
    // instead of ~13 (vector instructions are paid in full next to the MFMAs on this hardware).
    // (in REGISTERS: an LDS table read behind an LDS-DMA request makes the compiler wait for every memory operation in flight — it cannot tell the table from the
    // DMA's destination — which drained the stores the counted wait at the top of a unit leaves in flight)
    int rel[7];
    int ok_all = 0, ok_top = 0, ok_bottom = 0, ok_left = 0, ok_right = 0;                      // bit j: the chunk exists / is inside the image when the unit touches that border
    {
        // rows / columns a unit may read: top: patch row 0 is image row -1; bottom: the last unit row starts at image row 4 (UH - 1) - 1; likewise left / right
        const int ylast = p.H - (4 * (p.UH - 1) - 1), xlast = p.W - (16 * (p.UW - 1) - 1);      // patch rows / columns below these are inside the image
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int id = 64 * j + lane, px = id >> 2;
            const int pr = px / 18, pc = px - pr * 18;
            const int sq = (id & 3) ^ ((pc >> 2) & 3);
            rel[j] = ((pr * p.W + pc) * 32 + sq * 4) * 4;
            ok_all |= (px < 108 ? 1 : 0) << j;
            ok_top |= (pr >= 1 ? 1 : 0) << j;
            ok_bottom |= (pr < ylast ? 1 : 0) << j;
            ok_left |= (pc >= 1 ? 1 : 0) << j;
            ok_right |= (pc < xlast ? 1 : 0) << j;
        }
    }
    __syncthreads();
    // input transform: lane (tile t = lane % 16 at (t & 7, t >> 3) of the 8 x 2 patch, channel k4 = lane / 16 of the K-step's four).  LDS word of pixel
    // (2 tyl + i, 2 txl + j), K-step s: base + (i * 18 + j) * 16 + ((swizzle of the pixel's column ^ s) << 2): two swizzles per lane (columns j < 2 / j >= 2)
    const int t16 = lane & 15, k4 = lane >> 4;
    const int txl = t16 & 7, tyl = t16 >> 3;
    const int d_base = (2 * tyl * 18 + 2 * txl) * 16 + k4;
    const int swz_a = (txl >> 1) & 3, swz_b = ((txl + 1) >> 1) & 3;
    // write-out: lane (cout n = lane % 16 of each column block, tiles 4 (lane / 16) + i)
    const int n16 = lane & 15, r4 = lane >> 4;
    const float lo = p.relu ? 0.f : -INFINITY;
    const f32x4* Ul = reinterpret_cast<const f32x4*>(Us) + lane;         // + ((S * 2 + nb) * 4 + quarter) * 64: a quarter's 64 lanes are one contiguous KB (no bank conflicts)

    const int gw = blockIdx.x * C32_WAVES + wave, GW = gridDim.x * C32_WAVES;

    auto stage = [&](int unit, int half) {                // the 16-channel half of the unit's patch -> this wave's buffer (zeros outside the image = the padding)
        const int img = fdiv(unit, p.d_upi), rem = unit - img * p.upi;
        const int uy = fdiv(rem, p.d_uw), ux = rem - uy * p.UW;
        const int okbits = ok_all & (uy == 0 ? ok_top : -1) & (uy == p.UH - 1 ? ok_bottom : -1) & (ux == 0 ? ok_left : -1) & (ux == p.UW - 1 ? ok_right : -1);
        const int origin = (((img * p.H + 4 * uy - 1) * p.W + 16 * ux - 1) * 32 + half * 16) * 4;        // byte offset of patch pixel (0, 0), this half (may be negative: never used then)
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const unsigned off = (okbits >> j) & 1 ? (unsigned)(origin + rel[j]) : C32_OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (c32_lds_void*)(Ps + j * 256), 16, off, 0, 0, 0);
        }
    };

    int unit = gw;
    if (unit < p.units) stage(unit, 0);
    bool first = true;
    for (; unit < p.units; unit += GW) {
        f32x4 acc[16][2];                                 // (written, not accumulated, by the first K-step: no 128 zeroing moves per unit)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // this half's patch: requested before the write-out of the unit before (half 0: its 32 stores are younger and stay in flight) / inside the K-steps above (half 1)
            if (half == 0 && !first) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            float d[16];
            auto load_d = [&](int s) {
                const float* pa = Ps + d_base + ((swz_a ^ s) << 2);
                const float* pb = Ps + d_base + ((swz_b ^ s) << 2);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) d[i * 4 + j] = (j < 2 ? pa : pb)[(i * 18 + j) * 16];
            };
            load_d(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // V = B^T d B on register PAIRS (packed fp32: two values per instruction; vector instructions are paid in full next to the MFMAs): rows as plain
                // pair arithmetic, columns with operand selects: (v0, v1) = (t0 - t2, t1 + t2), (v2, v3) = (t2 - t1, t1 - t3) from A = (t0, t1), B = (t2, t3)
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
                const int S = half * 4 + s;
                f32x4 ufc[4], uf1[4];
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) ufc[qd] = Ul[((S * 2 + 0) * 4 + qd) * 64];
                // the next K-step's pixels are requested now and arrive behind this step's 32 MFMAs; after the half's last reads the buffer goes to the next request
                if (s < 3) load_d(s + 1);
                else {
                    if (half == 0) stage(unit, 1);
                    else if (unit + GW < p.units) stage(unit + GW, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) uf1[qd] = Ul[((S * 2 + 1) * 4 + qd) * 64];  // (behind the first 16 MFMAs)
                const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[q], ufc[q >> 2][q & 3], (half == 0 && s == 0) ? zero4 : acc[q][0], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[q], uf1[q >> 2][q & 3], (half == 0 && s == 0) ? zero4 : acc[q][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);        // (nothing else hoisted over the MFMAs: 128 accumulator registers leave no room)
            }
        }
        first = false;
        // ---- output transform (registers only), then all skip loads, then all stores: tile 4 r4 + i, cout 16 nb + n16 ----
        const int img = fdiv(unit, p.d_upi), rem = unit - img * p.upi;
        const int uy = fdiv(rem, p.d_uw), ux = rem - uy * p.UW;
        f32x4 yq[2][4];                                    // [column block][output pixel ab] over the lane's four tiles (vector arithmetic: packed adds, no register shuffles)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) {
            const float scn = SBs[16 * nb + n16], bin = SBs[32 + 16 * nb + n16];
            f32x4 t0[4], t1[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                t0[nu] = acc[0 * 4 + nu][nb] + acc[1 * 4 + nu][nb] + acc[2 * 4 + nu][nb];
                t1[nu] = acc[1 * 4 + nu][nb] - acc[2 * 4 + nu][nb] - acc[3 * 4 + nu][nb];
            }
            yq[nb][0] = (t0[0] + t0[1] + t0[2]) * scn + bin;
            yq[nb][1] = (t0[1] - t0[2] - t0[3]) * scn + bin;
            yq[nb][2] = (t1[0] + t1[1] + t1[2]) * scn + bin;
            yq[nb][3] = (t1[1] - t1[2] - t1[3]) * scn + bin;
        }
        const unsigned rowb = (unsigned)(p.W * 32 * 4);
        unsigned base[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tl = 4 * r4 + i;
            const int ty = 2 * uy + (tl >> 3), tx = 8 * ux + (tl & 7);
            base[i] = (ty < p.TH && tx < p.TW) ? (unsigned)((((img * p.H + 2 * ty) * p.W + 2 * tx) * 32 + n16) * 4) : C32_OOB;
        }
        if (p.res) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int ab = 0; ab < 4; ++ab)
                        yq[nb][ab][i] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, base[i] != C32_OOB ? base[i] + (ab >> 1) * rowb + (ab & 1) * 128u + nb * 64u : C32_OOB, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int ab = 0; ab < 4; ++ab)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(yq[nb][ab][i], lo)), yr, base[i] != C32_OOB ? base[i] + (ab >> 1) * rowb + (ab & 1) * 128u + nb * 64u : C32_OOB, 0, 0);
    }
}

// (Cout = 32, Cin = 32, 3, 3) -> U = G g G^T in the kernel's LDS order: [K-step S][column block nb][xi][lane = n + 16 k][nu] (position p = 4 xi + nu), channel 4 S + k, cout 16 nb + n
__global__ __launch_bounds__(256) void winograd_c32_pack_kernel(const float* __restrict__ w, float* __restrict__ u) {
    const int id = blockIdx.x * 256 + threadIdx.x;        // (S, nb, lane)
    if (id >= 8 * 2 * 64) return;
    const int lane = id & 63, nb = (id >> 6) & 1, S = id >> 7;
    const int c = 4 * S + (lane >> 4), n = 16 * nb + (lane & 15);
    const float* g = w + ((long long)n * 32 + c) * 9;
    float gg[4][3];                                       // G g: rows (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        gg[0][j] = g[0 * 3 + j];
        gg[1][j] = 0.5f * (g[0 * 3 + j] + g[1 * 3 + j] + g[2 * 3 + j]);
        gg[2][j] = 0.5f * (g[0 * 3 + j] - g[1 * 3 + j] + g[2 * 3 + j]);
        gg[3][j] = g[2 * 3 + j];
    }
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {                      // quarter xi of (S, nb): [lane][nu]
        float* o = u + ((((S * 2 + nb) * 4 + xi) * 64 + lane) * 4);
        o[0] = gg[xi][0];
        o[1] = 0.5f * (gg[xi][0] + gg[xi][1] + gg[xi][2]);
        o[2] = 0.5f * (gg[xi][0] - gg[xi][1] + gg[xi][2]);
        o[3] = gg[xi][2];
    }
}

}  // namespace vatl

using namespace vatl;

extern "C" int64_t vatl_winograd_c32_weight_floats(void) { return C32_U_FLOATS; }

extern "C" int vatl_pack_winograd_c32_weight(const float* w, float* u, void* stream) {
    if (!w || !u) return fail(VATL_EINVAL, "pack_winograd_c32_weight: null pointer");
    hipLaunchKernelGGL(winograd_c32_pack_kernel, dim3(4), dim3(256), 0, (hipStream_t)stream, w, u);
    return check_launch("winograd_c32_pack");
}

extern "C" int vatl_conv3x3_winograd_c32_supported(int N, int H, int W, int Cin, int Cout) {
    return Cin == 32 && Cout == 32 && N > 0 && H >= 2 && W >= 2 && (H & 1) == 0 && (W & 1) == 0 && (long long)N * H * W * 32 < (1LL << 30) ? 1 : 0;
}

extern "C" int vatl_conv3x3_winograd_c32_fwd(const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y, int N, int H, int W,
                                             int relu, void* stream) {
    if (!x || !u || !y) return fail(VATL_EINVAL, "conv3x3_winograd_c32_fwd: null pointer");
    if (!vatl_conv3x3_winograd_c32_supported(N, H, W, 32, 32)) return fail(VATL_EINVAL, "conv3x3_winograd_c32_fwd: serves 32 -> 32 channels, even H and W, N * H * W * 32 < 2^30");
    C32Params p{};
    p.x = x; p.u = u; p.scale = scale; p.bias = bias; p.res = residual; p.y = y;
    p.N = N; p.H = H; p.W = W; p.TH = H / 2; p.TW = W / 2; p.UH = (p.TH + 1) / 2; p.UW = (p.TW + 7) / 8; p.relu = relu;
    p.upi = p.UH * p.UW; p.units = N * p.upi;
    p.bytes = (unsigned)((long long)N * H * W * 32 * 4);
    p.d_upi = make_fastdiv((unsigned)p.upi); p.d_uw = make_fastdiv((unsigned)p.UW);
    const int smem = C32_LDS_FLOATS * (int)sizeof(float);
    static std::atomic<unsigned> configured{0};
    auto kern = winograd_c32_kernel;
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, configured, "winograd_c32")) return rc;
    int grid = (p.units + C32_WAVES - 1) / C32_WAVES;
    if (grid > 256) grid = 256;                            // one block per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, (hipStream_t)stream, p);
    // executed MFMA FLOPs: units x 16 tiles x 32 cout x 32 cin x 16 positions
    meter_add(1, 2.0 * (double)p.units * 16.0 * 32.0 * 32.0 * 16.0);
    meter_route(kRouteWinoC32);
    return check_launch("winograd_c32");
}
