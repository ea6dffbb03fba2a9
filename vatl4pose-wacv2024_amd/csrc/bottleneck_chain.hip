// Tail of one bottleneck and head of the next in ONE persistent launch over 32-pixel tiles (ResNet stage 1, HRNet layer1: 64 -> 256 -> 64 channels):
//     T  = relu(scale3 * (A x W3^T) + bias3 + skip)        A: conv2 output (M, 64),  W3: 64 -> 256,  skip / T: (M, 256)        [conv3 + bn3 + add + relu, Resnet.py:120-128]
//     Y1 = relu(scale1 * (T x W1^T) + bias1)               W1: 256 -> 64                                                        [next block's conv1 + bn1 + relu, :104-108]
// T is stored (it is the next block's skip connection) but never re-read from HBM: the second GEMM takes it from LDS.  Each wave owns 64 of T's 256 channels in the first GEMM and
// exactly that K range of the second one (split K over the four waves, partial tiles summed in wave order through LDS: fixed order, batch-independent bits).
// Both filters stay in registers (64 + 64 values per lane); the A rows arrive by LDS-DMA one tile ahead, the skip rows are requested before the first GEMM and consumed after it,
// and the stores of a tile stay in flight across the barrier into the next one (counted s_waitcnt: VMEM operations retire in order).
// T has the bits of the tiled implicit GEMM (same k order); Y1 sums K in 4 x 64 pieces, so it differs from the two-launch path in the last bits (5e-7 of the plane maximum).
// Measured at 1024 crops of 64x48 pixels (tools/chain_bench.py): 2.00 ms against 2.48 - 2.64 ms for the two tiled launches; the first GEMM alone (w1 == NULL) 1.36 against 1.51 ms.
#include "common.h"

namespace vatl {

struct ChainParams {
    const float* a;          // (M, 64)
    const float* w3;         // packed [256][64]
    const float* scale3;
    const float* bias3;
    const float* res;        // (M, 256) or null
    float* t;                // (M, 256)
    const float* w1;         // packed [N2][256], or null: first GEMM only
    const float* scale1;
    const float* bias1;
    float* out2;             // (M, N2)
    int M, N2, relu3;
    int m_tiles;
    unsigned a_bytes, t_bytes, o_bytes, w3_bytes, w1_bytes;
};

constexpr unsigned BOOB = 0xFFFFFFFFu;
typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void blds_void;
__device__ __forceinline__ f32x4 bbuf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); }
__device__ __forceinline__ void bbuf_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(bu32x4, v), r, off, 0, 0); }

constexpr int BT_LDT = 260;                       // T tile row pitch (floats)
constexpr int BT_AS = 32 * 64;                    // one A stage: 32 rows x 64 floats, 16-byte chunks XOR-swizzled by (row & 7)
constexpr int BT_FLOATS = 2 * BT_AS + 4 * 32 * 68;      // the T tile (32 x 260 = 8320 floats) and, after it, the four partial tiles (4 x 32 x 68 = 8704) share the region

template <bool SECOND>
__global__ __launch_bounds__(256, 2) void bottleneck_chain_kernel(ChainParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                             // [2][32][64]
    float* Ts = smem + 2 * BT_AS;                 // [32][260]; after the second GEMM's reads: partial tiles [4][32][64 + 4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, h = lane >> 5;
    const int nblk = gridDim.x, bid = blockIdx.x;

    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t w3r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, p.w3_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t tr = __builtin_amdgcn_make_buffer_rsrc(p.t, 0, p.t_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.t), 0, p.res ? p.t_bytes : 0u, 0x00020000);

    // filters in MFMA B-fragment order: lane (channel fr of the 32-column block, k half h) holds W[n][8 g + 4 h .. + 3]
    f32x4 w3f[2][8];                              // first GEMM: this wave's 64 output channels, K = 64
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 8; ++g) w3f[nb][g] = bbuf_load4(w3r, (unsigned)(((wave * 64 + nb * 32 + fr) * 64 + 8 * g + 4 * h) * 4));
    float sc3[2], bi3[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int n = wave * 64 + nb * 32 + fr;
        sc3[nb] = p.scale3 ? p.scale3[n] : 1.f;
        bi3[nb] = p.bias3 ? p.bias3[n] : 0.f;
    }
    f32x4 w1f[SECOND ? 2 : 1][8];                 // second GEMM: all 64 output channels, this wave's K range [64 wave, 64 wave + 64)
    if constexpr (SECOND) {
        const __amdgpu_buffer_rsrc_t w1r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w1), 0, p.w1_bytes, 0x00020000);
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int g = 0; g < 8; ++g) w1f[nb][g] = bbuf_load4(w1r, (unsigned)(((nb * 32 + fr) * 256 + wave * 64 + 8 * g + 4 * h) * 4));
    }
    const float lo3 = p.relu3 ? 0.f : -INFINITY;

    // A stage by LDS-DMA: 512 16-byte pieces per tile = 2 wave instructions per wave; LDS piece q = (row q >> 4, position q & 15) receives the row's chunk (q & 15) ^ (row & 7)
    // (rows past M need no test anywhere below: their byte offsets lie past the descriptors' sizes, so the hardware drops those loads and stores)
    auto a_dma = [&](int buf, int mt) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = (wave * 2 + u) * 64 + lane;
            const int row = q >> 4, chunk = (q & 15) ^ (row & 7);
            const unsigned off = (unsigned)mt * (32u * 64u * 4u) + (unsigned)(row * 64 + chunk * 4) * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, (blds_void*)(As + buf * BT_AS + (wave * 2 + u) * 256), 16, off, 0, 0, 0);
        }
    };

    const int c4 = tid & 63, r0 = tid >> 6;
    const unsigned tlane = (unsigned)(r0 * 256 + c4 * 4) * 4u;
    int mt = bid, buf = 0;
    if (mt < p.m_tiles) a_dma(0, mt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; mt < p.m_tiles; mt += nblk, buf ^= 1) {
        __syncthreads();                          // this tile's A rows have landed (waited for at the end of the pass before); T of the tile before has been consumed
        // the skip-connection rows of this tile: requested now, consumed after the first GEMM (their latency behind 64 MFMAs per wave)
        const unsigned tbase = (unsigned)mt * (32u * 1024u) + tlane;
        f32x4 rs[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) rs[u] = p.res ? bbuf_load4(rr, tbase + (unsigned)u * 4096u) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (mt + nblk < p.m_tiles) a_dma(buf ^ 1, mt + nblk);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[2];
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(As + buf * BT_AS + fr * 64 + (((2 * g + h) ^ (fr & 7)) << 2));
#pragma unroll
            for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tt], w3f[nb][g][tt], acc[nb], 0, 0, 0);
        }
        // conv3 + bn3 into the T tile
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                Ts[row * BT_LDT + wave * 64 + nb * 32 + fr] = acc[nb][e] * sc3[nb] + bi3[nb];
            }
        __syncthreads();
        // + skip, ReLU, stored as the block output (16-byte rows) and written back to the tile for the second GEMM
        {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = r0 + 4 * u;
                const f32x4 v = *reinterpret_cast<const f32x4*>(&Ts[row * BT_LDT + c4 * 4]);
                f32x4 o;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[c] = fmaxf(v[c] + rs[u][c], lo3);
                bbuf_store4(tr, tbase + (unsigned)u * 4096u, o);
                if constexpr (SECOND) *reinterpret_cast<f32x4*>(&Ts[row * BT_LDT + c4 * 4]) = o;
            }
        }
        if constexpr (SECOND) {
            __syncthreads();
            // second GEMM over this wave's K range: A fragments from the T tile (lane = (pixel row fr, k half h))
            f32x16 acc2[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[nb][e] = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const f32x4 tf = *reinterpret_cast<const f32x4*>(&Ts[fr * BT_LDT + wave * 64 + 8 * g + 4 * h]);
#pragma unroll
                for (int tt = 0; tt < 4; ++tt)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x2f32(tf[tt], w1f[nb][g][tt], acc2[nb], 0, 0, 0);
            }
            __syncthreads();                      // every wave has read its K range of T: the tile region is free for the partial sums
            float* Pw = Ts + wave * (32 * 68);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
                    Pw[row * 68 + nb * 32 + fr] = acc2[nb][e];
                }
            __syncthreads();
            {
                const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(p.out2, 0, p.o_bytes, 0x00020000);
                const int q4 = tid & 15, q0 = tid >> 4;              // 16 quads x 16 rows per pass, 2 passes
                const f32x4 s1 = p.scale1 ? *reinterpret_cast<const f32x4*>(p.scale1 + q4 * 4) : f32x4{1.f, 1.f, 1.f, 1.f};
                const f32x4 b1 = p.bias1 ? *reinterpret_cast<const f32x4*>(p.bias1 + q4 * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int row = q0 + 16 * u;
                    f32x4 sum = *reinterpret_cast<const f32x4*>(&Ts[row * 68 + q4 * 4]);
#pragma unroll
                    for (int w = 1; w < 4; ++w) {
                        const f32x4 q = *reinterpret_cast<const f32x4*>(&Ts[w * (32 * 68) + row * 68 + q4 * 4]);
#pragma unroll
                        for (int c = 0; c < 4; ++c) sum[c] += q[c];
                    }
                    f32x4 o;
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = fmaxf(sum[c] * s1[c] + b1[c], 0.f);
                    bbuf_store4(orr, ((unsigned)(mt * 32 + row) * 64u + (unsigned)q4 * 4u) * 4u, o);
                }
            }
        }
        // the next tile's A rows were requested before this pass's stores: wait for them only (vmcnt retires in order), the stores stay in flight across the barrier
        if constexpr (SECOND) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
}

}  // namespace vatl

using namespace vatl;

static bool chain_shape_ok(int Cmid, int Cout, int Cnext, int64_t M) {
    return Cmid == 64 && Cout == 256 && (Cnext == 64 || Cnext == 0) && M > 0 && (M + 32) * 256 < (1LL << 30);       // 32-bit byte offsets into T, tile rounding included
}

extern "C" int vatl_bottleneck_chain_supported(int Cmid, int Cout, int Cnext, int64_t M) { return chain_shape_ok(Cmid, Cout, Cnext, M) ? 1 : 0; }

extern "C" int vatl_bottleneck_chain_fwd(const float* a, const float* w3, const float* scale3, const float* bias3, const float* skip, float* t, const float* w1,
                                         const float* scale1, const float* bias1, float* y1, int64_t M, int Cmid, int Cout, int Cnext, void* stream) {
    if (!a || !w3 || !t || (Cnext != 0) != (w1 != nullptr) || (w1 && !y1)) return fail(VATL_EINVAL, "bottleneck_chain_fwd: bad arguments");
    if (!chain_shape_ok(Cmid, Cout, Cnext, M)) return fail(VATL_EINVAL, "bottleneck_chain_fwd: serves 64 -> 256 (-> 64) channels and (M + 32) * 256 < 2^30 pixels x channels");
    ChainParams p{};
    p.a = a; p.w3 = w3; p.scale3 = scale3; p.bias3 = bias3; p.res = skip; p.t = t; p.w1 = w1; p.scale1 = scale1; p.bias1 = bias1; p.out2 = y1;
    p.M = (int)M; p.N2 = 64; p.relu3 = 1; p.m_tiles = (int)((M + 31) / 32);
    p.a_bytes = (unsigned)(M * 64 * 4); p.t_bytes = (unsigned)(M * 256 * 4); p.o_bytes = (unsigned)(M * 64 * 4); p.w3_bytes = 256 * 64 * 4; p.w1_bytes = 64 * 256 * 4;
    const int smem = BT_FLOATS * (int)sizeof(float);
    static std::atomic<unsigned> c0{0}, c1{0};
    const double padded = (double)p.m_tiles * 32.0;
    if (w1) {
        auto kern = bottleneck_chain_kernel<true>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, c1, "bottleneck_chain")) return rc;
        const int grid = p.m_tiles < 512 ? p.m_tiles : 512;                   // 237 registers: two blocks per CU
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, (hipStream_t)stream, p);
        meter_add(0, 2.0 * padded * 256.0 * 64.0 * 2.0);
        meter_route(kRouteChain);
    } else {
        auto kern = bottleneck_chain_kernel<false>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, c0, "bottleneck_chain")) return rc;
        const int grid = p.m_tiles < 512 ? p.m_tiles : 512;                   // (163 registers: a third block per CU fits and was measured: no gain)
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, (hipStream_t)stream, p);
        meter_add(0, 2.0 * padded * 256.0 * 64.0);
        meter_route(kRouteChain);
    }
    return check_launch("bottleneck_chain");
}
