// 1x1 convolution with K = 128 as a ROW-STREAMING GEMM (round 4): y = act(scale * (A x W^T) + bias + skip), A (M, 128) [or two (M, 64) halves: the
// dual-source form, conv3 + projection shortcut of ResNet's first block], W [N][128], N a multiple of 128.
// The tiled implicit GEMM runs these short-K / wide-N layers at 0.61 of either roof (l2.n.c3: 1.07 ms against 0.66 ms of MFMA time and 0.66 ms of HBM time): with
// four k-tiles per 128 x 128 tile its epilogue (LDS transpose, skip read, store) is as long as its k-loop and does not overlap it.  Here, as in
// csrc/bottleneck_chain.hip: a block keeps ONE 128-channel slice of the filter in registers for its whole life (64 values per lane: wave w owns channels 32 w .. 32 w + 31
// of the slice), walks 32-pixel tiles, the A rows arrive by LDS-DMA one tile ahead, the skip rows are requested BEFORE the tile's MFMAs and consumed after them, and the
// tile's stores stay in flight across the barrier into the next tile (counted s_waitcnt: VMEM operations retire in order).  Blocks that share pixel tiles (the N / 128
// slices) are neighbours on one XCD, so A is read from HBM once.  Same k order as the tiled kernels: bit-identical results.
#include "common.h"

#include <atomic>

namespace vatl {

struct RowsParams {
    const float* a;          // (M, 128), or (M, 64) when x2 is set
    const float* x2;         // (M, 64) second source (K columns 64 .. 127), or null
    const float* w;          // [N][128]
    const float* scale;      // (N) or null
    const float* bias;       // (N) or null
    const float* res;        // (M, N) or null
    float* y;                // (M, N)
    int M, N, relu, m_tiles, nslices;
    unsigned a_bytes, x2_bytes, y_bytes, w_bytes;
};

typedef unsigned int ru32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void rlds_void;
__device__ __forceinline__ f32x4 rbuf_load4(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)); }
__device__ __forceinline__ void rbuf_store4(__amdgpu_buffer_rsrc_t r, unsigned off, f32x4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ru32x4, v), r, off, 0, 0); }

constexpr int RW_AS = 32 * 128;                   // one A stage: 32 rows x 128 floats, 16-byte chunks XOR-swizzled by (row & 7)
constexpr int RW_LDT = 132;                       // output tile row pitch (floats)
constexpr int RW_FLOATS = 2 * RW_AS + 32 * RW_LDT;

template <bool DUAL>
__global__ __launch_bounds__(256, 3) void conv1x1_rows_kernel(RowsParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                             // [2][32][128]
    float* Ts = smem + 2 * RW_AS;                 // [32][132]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, h = lane >> 5;
    // block -> (slice, first tile, tile step): the nslices blocks of a pixel tile are consecutive on one XCD (block b runs on XCD b % 8)
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int slice = loc % p.nslices, seq = loc / p.nslices;
    const int tstep = (gridDim.x >> 3) / p.nslices * 8;
    const int n0 = slice * 128;

    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DUAL ? p.x2 : p.a), 0, DUAL ? p.x2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0, p.res ? p.y_bytes : 0u, 0x00020000);

    // the filter slice in MFMA B-fragment order: lane (channel fr of the wave's 32, k half h) holds W[n][8 g + 4 h .. + 3], g = 0 .. 15
    f32x4 wf[16];
    const int nw = n0 + wave * 32 + fr;
#pragma unroll
    for (int g = 0; g < 16; ++g) wf[g] = rbuf_load4(wr, (unsigned)((nw * 128 + 8 * g + 4 * h) * 4));
    const float sc = p.scale ? p.scale[nw] : 1.f, bi = p.bias ? p.bias[nw] : 0.f;
    const float lo = p.relu ? 0.f : -INFINITY;

    // A stage by LDS-DMA: 1024 16-byte pieces per tile = 4 wave instructions per wave; LDS piece q = (row q >> 5, position q & 31) receives the row's chunk
    // (q & 31) ^ (row & 7).  Rows past M need no test: their byte offsets lie past the descriptors' sizes (loads give zeros, stores are dropped).
    auto a_dma = [&](int buf, int mt) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = (wave * 4 + u) * 64 + lane;
            const int row = q >> 5, chunk = (q & 31) ^ (row & 7);
            rlds_void* dst = (rlds_void*)(As + buf * RW_AS + (wave * 4 + u) * 256);
            if constexpr (DUAL) {                                   // chunks 0 .. 15 from a (M, 64), 16 .. 31 from x2 (M, 64): two requests with complementary lanes
                const unsigned off = (unsigned)mt * (32u * 64u * 4u) + (unsigned)(row * 64 + (chunk & 15) * 4) * 4u;
                if (chunk < 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, dst, 16, off, 0, 0, 0);
                else            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, dst, 16, off, 0, 0, 0);
            } else {
                const unsigned off = (unsigned)mt * (32u * 128u * 4u) + (unsigned)(row * 128 + chunk * 4) * 4u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, dst, 16, off, 0, 0, 0);
            }
        }
    };

    const int c4 = tid & 31, r0 = tid >> 5;       // write-out: 32 float4 columns x 8 rows per pass, 4 passes
    const unsigned nrow = (unsigned)p.N * 4u;     // bytes of an output row
    const unsigned ylane = (unsigned)(r0 * p.N + n0 + c4 * 4) * 4u;
    int mt = seq * 8 + xcd, buf = 0;
    if (mt < p.m_tiles) a_dma(0, mt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; mt < p.m_tiles; mt += tstep, buf ^= 1) {
        __syncthreads();                          // this tile's A rows have landed (waited for at the end of the pass before); the output tile of the pass before has been consumed
        const unsigned ybase = (unsigned)mt * (32u * nrow) + ylane;
        f32x4 rs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) rs[u] = p.res ? rbuf_load4(rr, ybase + (unsigned)u * (8u * nrow)) : f32x4{0.f, 0.f, 0.f, 0.f};
        if (mt + tstep < p.m_tiles) a_dma(buf ^ 1, mt + tstep);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(As + buf * RW_AS + fr * 128 + (((2 * g + h) ^ (fr & 7)) << 2));
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tt], wf[g][tt], acc, 0, 0, 0);
        }
        // (stores straight from the accumulator layout — 16 four-byte stores of 128-byte runs, no LDS tile, one barrier per tile — measured equal at three
        // blocks per CU and slower at four, where the kernel spills: profiles/r04_notes.md)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
            Ts[row * RW_LDT + wave * 32 + fr] = acc[e] * sc + bi;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int row = r0 + 8 * u;
            const f32x4 v = *reinterpret_cast<const f32x4*>(&Ts[row * RW_LDT + c4 * 4]);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) o[c] = fmaxf(v[c] + rs[u][c], lo);
            rbuf_store4(yr, ybase + (unsigned)u * (8u * nrow), o);
        }
        // the next tile's A rows were requested before this pass's four stores: wait for them only, the stores stay in flight across the barrier
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
}

// K = 256 (round 6: Bottleneck.conv3 of ResNet stage 3, 256 -> 1024 + skip, five launches of the headline step at 0.74 of their MFMA time on the tiled kernel, whose
// 128 x 128 tile has one write-out per eight k-tiles).  Same scheme with the numbers doubled: 128 filter values per lane, A stages of 32 rows x 256 floats (two of them =
// 64 KB: two blocks per CU), eight DMA instructions per wave and tile, 128 MFMAs per wave and tile.  The output tile does NOT go through LDS (no room for it next to a
// second block, and the K = 128 kernel measured the two forms equal): a lane's 16 accumulator values are 16 rows of ONE channel, the 32 lanes of a half wave store a
// 128-byte run per row; the skip values are read the same way before the MFMAs.  One barrier per tile.  Same k order as the tiled kernels: bit-identical.
constexpr int RW2_K = 256;
constexpr int RW2_AS = 32 * RW2_K;
constexpr int RW2_FLOATS = 2 * RW2_AS;

__global__ __launch_bounds__(256, 2) void conv1x1_rows256_kernel(RowsParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                             // [2][32][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 31, h = lane >> 5;
    const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
    const int slice = loc % p.nslices, seq = loc / p.nslices;
    const int tstep = (gridDim.x >> 3) / p.nslices * 8;
    const int n0 = slice * 128;

    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.a), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.y_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0, p.res ? p.y_bytes : 0u, 0x00020000);

    f32x4 wf[32];                                 // lane (channel fr of the wave's 32, k half h): W[n][8 g + 4 h .. + 3], g = 0 .. 31
    const int nw = n0 + wave * 32 + fr;
#pragma unroll
    for (int g = 0; g < 32; ++g) wf[g] = rbuf_load4(wr, (unsigned)((nw * RW2_K + 8 * g + 4 * h) * 4));
    const float sc = p.scale ? p.scale[nw] : 1.f, bi = p.bias ? p.bias[nw] : 0.f;
    const float lo = p.relu ? 0.f : -INFINITY;

    // A stage: 2048 16-byte pieces per tile = 8 wave instructions per wave; LDS piece q = (row q >> 6, position q & 63) receives the row's chunk (q & 63) ^ (row & 7)
    auto a_dma = [&](int buf, int mt) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int q = (wave * 8 + u) * 64 + lane;
            const int row = q >> 6, chunk = (q & 63) ^ (row & 7);
            rlds_void* dst = (rlds_void*)(As + buf * RW2_AS + (wave * 8 + u) * 256);
            const unsigned off = (unsigned)mt * (32u * RW2_K * 4u) + (unsigned)(row * RW2_K + chunk * 4) * 4u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, dst, 16, off, 0, 0, 0);
        }
    };

    const unsigned nrow = (unsigned)p.N * 4u;     // bytes of an output row
    const unsigned ylane = (unsigned)((4 * h) * p.N + nw) * 4u;       // accumulator element e: row (e & 3) + 8 (e >> 2) + 4 h, channel nw
    int mt = seq * 8 + xcd, buf = 0;
    if (mt < p.m_tiles) a_dma(0, mt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (; mt < p.m_tiles; mt += tstep, buf ^= 1) {
        __syncthreads();                          // this tile's A rows have landed (waited for at the end of the pass before); every wave is done reading the other buffer
        const unsigned ybase = (unsigned)mt * (32u * nrow) + ylane;
        float rs[16];
#pragma unroll
        for (int e = 0; e < 16; ++e)
            rs[e] = p.res ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, ybase + (unsigned)((e & 3) + 8 * (e >> 2)) * nrow, 0, 0)) : 0.f;
        if (mt + tstep < p.m_tiles) a_dma(buf ^ 1, mt + tstep);
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(As + buf * RW2_AS + fr * RW2_K + (((2 * g + h) ^ (fr & 7)) << 2));
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[tt], wf[g][tt], acc, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = fmaxf(acc[e] * sc + bi + rs[e], lo);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, ybase + (unsigned)((e & 3) + 8 * (e >> 2)) * nrow, 0, 0);
        }
        // the next tile's A rows were requested before this pass's 16 stores: wait for them only, the stores stay in flight across the barrier
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
}

}  // namespace vatl

using namespace vatl;

static bool rows_shape_ok(int K1, int K2, int N, long long M) {
    return ((K1 == 128 && K2 == 0) || (K1 == 64 && K2 == 64) || (K1 == 256 && K2 == 0)) && N >= 128 && N % 128 == 0 && N <= 4096 && M > 0 && (M + 32) * (long long)N < (1LL << 30)
        && (M + 32) * (long long)(K1 + K2) < (1LL << 30);
}

extern "C" int vatl_conv1x1_rows_supported(int K1, int K2, int N, int64_t M) { return rows_shape_ok(K1, K2, N, M) ? 1 : 0; }

extern "C" int vatl_conv1x1_rows_fwd(const float* a, const float* x2, const float* w, const float* scale, const float* bias, const float* residual, float* y,
                                     int64_t M, int K1, int K2, int N, int relu, void* stream) {
    if (!a || !w || !y || (K2 != 0) != (x2 != nullptr)) return fail(VATL_EINVAL, "conv1x1_rows_fwd: bad arguments");
    if (!rows_shape_ok(K1, K2, N, M))
        return fail(VATL_EINVAL, "conv1x1_rows_fwd: serves K = 128 (or 64 + 64 from two tensors) and K = 256, N a multiple of 128 up to 4096, (M + 32) * N < 2^30");
    RowsParams p{};
    p.a = a; p.x2 = x2; p.w = w; p.scale = scale; p.bias = bias; p.res = residual; p.y = y;
    p.M = (int)M; p.N = N; p.relu = relu; p.m_tiles = (int)((M + 31) / 32); p.nslices = N / 128;
    p.a_bytes = (unsigned)(M * K1 * 4); p.x2_bytes = (unsigned)(M * K2 * 4); p.y_bytes = (unsigned)(M * N * 4); p.w_bytes = (unsigned)((long long)N * (K1 + K2) * 4);
    if (K1 == 256) {                              // two blocks per CU; whole tile teams per XCD like below
        int teams = 512 / (8 * p.nslices);
        if (teams < 1) teams = 1;
        const int need = (p.m_tiles + 7) / 8;
        if (teams > need) teams = need;
        const int grid = teams * 8 * p.nslices;
        const int smem2 = RW2_FLOATS * (int)sizeof(float);
        static std::atomic<unsigned> c2{0};
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv1x1_rows256_kernel), smem2, c2, "conv1x1_rows256")) return rc;
        hipLaunchKernelGGL(conv1x1_rows256_kernel, dim3(grid), dim3(256), smem2, (hipStream_t)stream, p);
        meter_add(0, 2.0 * ((double)p.m_tiles * 32.0) * (double)N * 256.0);
        meter_route(kRouteRows);
        return check_launch("conv1x1_rows256");
    }
    const int smem = RW_FLOATS * (int)sizeof(float);
    // three blocks per CU; the grid is a multiple of 8 x nslices (whole tile teams per XCD), never more teams than tiles
    int teams = 768 / (8 * p.nslices);
    if (teams < 1) teams = 1;
    const int need = (p.m_tiles + 7) / 8;
    if (teams > need) teams = need;
    const int grid = teams * 8 * p.nslices;
    static std::atomic<unsigned> c0{0}, c1{0};
    if (x2) {
        auto kern = conv1x1_rows_kernel<true>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, c1, "conv1x1_rows")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, (hipStream_t)stream, p);
    } else {
        auto kern = conv1x1_rows_kernel<false>;
        if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, c0, "conv1x1_rows")) return rc;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, (hipStream_t)stream, p);
    }
    meter_add(0, 2.0 * ((double)p.m_tiles * 32.0) * (double)N * 128.0);
    meter_route(kRouteRows);
    return check_launch("conv1x1_rows");
}
