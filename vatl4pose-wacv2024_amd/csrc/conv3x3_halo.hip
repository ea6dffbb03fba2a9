// 3x3 / stride 1 / pad 1 convolution for NARROW layers (Cin = Cout = 32: the high-resolution branch of HRNet-W32,
// hrnet.py:24-56 BasicBlock — 64 of the network's 293 convolutions and 31 % of its conv time on the generic kernel).
//
// Why a second kernel.  The generic implicit GEMM (conv_igemm.hip) gives such a layer a 128 x 32 tile and nine k-tiles per
// block: every block re-gathers its 128 input pixels nine times through L2 (once per filter tap: 147 KB of operand loads for
// 16 KB of output), re-loads the 36 KB filter, and pays a prologue (first loads with nothing to overlap) and an epilogue as
// long as its 144 MFMAs per wave — 65 % MFMA-busy at best (profiles/r01_notes.md).  Here a block is PERSISTENT and
//   * keeps the whole filter [32][9*32] in LDS for its lifetime (36.5 KB, rows padded to 292 floats: conflict-free
//     ds_read_b128 fragment reads),
//   * loads each input pixel ONCE: an output patch of TH x TW = 128 pixels is served by its (TH+2) x (TW+2) halo tile in LDS
//     (pixel stride 36 floats, zero rows outside the image through buffer bounds checks), and the nine taps are nine shifted
//     views of that tile — the MFMA A fragments are read straight from it,
//   * requests the NEXT patch's halo (6 x 16 bytes per thread) before the 144 MFMAs of the current one and stores it after
//     them: no prologue, two barriers per patch, no per-k-tile barrier,
//   * writes the output from the accumulators directly (a 32x32 MFMA tile has one lane per channel: a store instruction writes
//     two full 128-byte NHWC pixel rows), scale / bias / residual / ReLU fused.
// Measured and dropped (MI355X, hr.b32 at 1024 crops: this kernel 471 us, implicit GEMM 579-600 us):
//   * two or three 4-wave groups per block sharing one filter copy (8 / 12 waves per CU behind one barrier): 505 / 495 us;
//   * a 64-channel variant — halo tile resident, the 147 KB filter streamed one tap at a time through two LDS buffers in a ring
//     that runs across patches: bit-identical, hr.b64 577 us against 480 us on the implicit GEMM, l1.c2 equal (ten barriers per
//     patch and the first fragment reads of every tap exposed: the generic kernel's structure and its 79 %).
// Reduction order per output = (tap, 8-channel group, pair) exactly as conv_igemm_kernel's k-tiles: results are BIT-IDENTICAL to
// the generic kernel (tests/test_gpu_conv.py), so the evaluation path keeps its batch-position-independent bits.
#include "common.h"

namespace vatl {

struct HaloParams {
    const float* x;
    const float* w;        // packed [32][3][3][32] (vatl_pack_conv_weight)
    const float* scale;
    const float* bias;
    const float* res;
    float* y;
    int N, H, W, relu;
    int ppx, ppy, total;   // patches per image row / column, patches in the launch
    unsigned x_bytes;      // extent of x and of y / res (same shape)
};

constexpr unsigned HOOB = 0xFFFFFFFFu;
typedef unsigned int hu32x4 __attribute__((ext_vector_type(4)));

template <int C, int TH, int TW>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(HaloParams p) {
    constexpr int PS = C + 4;                  // halo pixel stride (floats)
    constexpr int KW = 9 * C + 4;              // filter row stride (floats)
    constexpr int HWD = TW + 2, HP = (TH + 2) * HWD;
    constexpr int C4 = C / 4;
    constexpr int NLD = (HP * C4 + 255) / 256; // 16-byte halo loads per thread
    constexpr int RPW = 32 / TW;               // patch rows owned by a wave
    static_assert(C == 32 && TH * TW == 128 && 32 % TW == 0 && TH == 4 * RPW, "four waves x 32 pixels x 32 channels");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;                          // [C][KW]
    float* Hs = smem + C * KW;                 // [HP][PS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 31, hi = lane >> 5;
    int t = blockIdx.x;
    if (t >= p.total) return;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res), 0, p.res ? p.x_bytes : 0u, 0x00020000);

    // ---- per-thread halo slots (the same for every patch): halo pixel, 16-byte chunk, LDS position
    int hy[NLD], hx[NLD], hq[NLD], hoff[NLD];
    bool hv[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 256 * i;
        hv[i] = idx < HP * C4;
        const int hp = idx / C4;
        hq[i] = (idx - hp * C4) * 4;
        hy[i] = hp / HWD;
        hx[i] = hp - hy[i] * HWD;
        hoff[i] = hp * PS + hq[i];
    }
    const int ppi = p.ppx * p.ppy;
    auto origin = [&](int tt, int& n, int& y0, int& x0) {
        n = tt / ppi;
        const int rem = tt - n * ppi;
        const int pyb = rem / p.ppx;
        y0 = pyb * TH;
        x0 = (rem - pyb * p.ppx) * TW;
    };
    auto issue = [&](int tt, f32x4 (&r)[NLD]) {        // halo of patch tt -> registers (zeros outside the image / past the end)
        int n, y0, x0;
        origin(tt, n, y0, x0);
        const bool live = tt < p.total;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int iy = y0 - 1 + hy[i], ix = x0 - 1 + hx[i];
            const bool ok = live && hv[i] && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            const unsigned off = ok ? (unsigned)(((n * p.H + iy) * p.W + ix) * C + hq[i]) << 2 : HOOB;
            r[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        }
    };
    auto stash = [&](const f32x4 (&r)[NLD]) {
#pragma unroll
        for (int i = 0; i < NLD; ++i)
            if (hv[i]) *reinterpret_cast<f32x4*>(&Hs[hoff[i]]) = r[i];
    };

    f32x4 hr[NLD];
    issue(t, hr);
    // the filter: C rows of 9*C floats, contiguous in the packed layout
    for (int i = tid; i < C * 9 * C4; i += 256) {
        const int n = i / (9 * C4), q = i - n * (9 * C4);
        *reinterpret_cast<f32x4*>(&Ws[n * KW + q * 4]) = reinterpret_cast<const f32x4*>(p.w)[i];
    }
    stash(hr);
    __syncthreads();

    // ---- fragment addressing: lane -> (pixel frow of this wave's 32, channel pair group hi)
    const int lpy = wave * RPW + frow / TW, lpx = frow % TW;
    const float* abase = Hs + (lpy * HWD + lpx) * PS + hi * 4;
    const float* bbase = Ws + frow * KW + hi * 4;
    const float sc = p.scale ? p.scale[frow] : 1.f;
    const float bi = p.bias ? p.bias[frow] : 0.f;
    const float lo = p.relu ? 0.f : -INFINITY;

    for (;;) {
        const int tn = t + gridDim.x;
        issue(tn, hr);                                     // next patch's halo: in flight during this patch's MFMAs
        int n, y0, x0;
        origin(t, n, y0, x0);
        // output offsets of this lane's 16 accumulator rows (pixels) and the residual values, requested up front
        unsigned ooff[16];
        float rv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int prow = (e & 3) + 8 * (e >> 2) + 4 * hi;
            const int oy = y0 + wave * RPW + prow / TW, ox = x0 + prow % TW;
            ooff[e] = (unsigned)(((n * p.H + oy) * p.W + ox) * C + frow) << 2;
        }
        if (p.res) {
#pragma unroll
            for (int e = 0; e < 16; ++e) rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, ooff[e], 0, 0));
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) rv[e] = 0.f;
        }
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* ap = abase + ((tap / 3) * HWD + (tap % 3)) * PS;
            const float* bp = bbase + tap * C;
#pragma unroll
            for (int g = 0; g < C / 8; ++g) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(ap + g * 8);
                const f32x4 bf = *reinterpret_cast<const f32x4*>(bp + g * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[k], bf[k], acc, 0, 0, 0);
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = acc[e] * sc + bi;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fmaxf(v + rv[e], lo)), yr, ooff[e], 0, 0);
        }
        __syncthreads();                                   // every wave is done reading this patch's halo
        if (tn >= p.total) break;
        stash(hr);
        __syncthreads();
        t = tn;
    }
}

static std::atomic<int> g_halo_on{1};

template <int TH, int TW>
static int launch_halo(HaloParams p, hipStream_t st) {
    auto kern = conv3x3_halo_kernel<32, TH, TW>;
    constexpr int smem = (32 * (9 * 32 + 4) + (TH + 2) * (TW + 2) * 36) * (int)sizeof(float);
    static std::atomic<unsigned> configured{0};
    if (int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), smem, configured, "conv3x3_halo")) return rc;
    p.ppx = p.W / TW;
    p.ppy = p.H / TH;
    p.total = p.N * p.ppx * p.ppy;
    const int grid = p.total < 512 ? p.total : 512;        // two resident blocks per CU, each walks total / 512 patches
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), smem, st, p);
    meter_add(0, 2.0 * (double)p.total * TH * TW * 32.0 * 32.0 * 9.0);
    meter_route(kRouteHalo);
    return check_launch("conv3x3_halo");
}

// Called by vatl_conv2d_fwd: returns 1 when the layer is not one of the shapes this kernel serves (the caller falls through to
// the generic implicit GEMM), else the launch status.
int conv3x3_halo_try(const float* x, const float* w, const float* scale, const float* bias, const float* residual, float* y, int N, int H, int W,
                     int Cin, int Cout, int CoutPad, int R, int S, int stride, int pad, int relu, hipStream_t st) {
    if (!g_halo_on.load(std::memory_order_relaxed)) return 1;
    if (Cin != 32 || Cout != 32 || CoutPad != 32 || R != 3 || S != 3 || stride != 1 || pad != 1) return 1;
    const long long elems = (long long)N * H * W * 32;
    if (elems >= (1LL << 30)) return 1;
    HaloParams p{};
    p.x = x; p.w = w; p.scale = scale; p.bias = bias; p.res = residual; p.y = y;
    p.N = N; p.H = H; p.W = W; p.relu = relu;
    p.x_bytes = (unsigned)(elems * 4);
    if (W % 16 == 0 && H % 8 == 0) return launch_halo<8, 16>(p, st);
    if (W % 8 == 0 && H % 16 == 0) return launch_halo<16, 8>(p, st);
    return 1;
}

int conv3x3_halo_enable(int on) { g_halo_on.store(on ? 1 : 0, std::memory_order_relaxed); return 0; }

}  // namespace vatl
