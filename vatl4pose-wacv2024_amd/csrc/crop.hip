// Crop producer — the step right before the backbone:
//   SimpleTransform.test_transform / __call__     alphapose/utils/presets/simple_transform.py:81-98, 179-251
//     cv2.warpAffine(img, trans, (w, h), flags=cv2.INTER_LINEAR)   -> fixed-point bilinear, BORDER_CONSTANT 0
//     im_to_torch (alphapose/utils/transforms.py:76-91)            -> HWC u8 -> CHW fp32, / 255 when the crop's max > 1
//     img[c].add_(-mean[c])                                        -> simple_transform.py:93-95
// The warp restates OpenCV 4.8's integer algorithm (imgwarp.cpp WarpAffineInvoker + remapBilinear<FixedPtCast<int,uchar,15>>):
// 10-bit fixed-point source coordinates rounded to 1/32 px, int16 weights scaled by 2^15.  All of it is integer work
// after the two double multiplies per row / column, so the u8 result is defined bit for bit.
// Write-dominated: 12 B per output pixel against <= 12 source bytes that mostly hit in L2; 2.86 TB/s measured.  What bounds it is the
// texture addresser (TA busy 93 % of the kernel, ~52 cycles per gather instruction; VALU 47 %: profiles/r03_notes.md) — aligned 12-byte
// loads + a funnel shift instead of the unaligned 8-byte ones changed nothing, lanes along the output row (2 - 4 cache lines per gather
// instead of ~15, but four times the store instructions) was 15 % slower.
#include "common.h"

#include <atomic>

namespace vatl {

struct CropParams {
    const uint8_t* arena;
    const long long* src_off;     // (B)   byte offset of the crop's frame
    const int* src_hwf;           // (B,3) frame height, width, mirror flag
    const double* minv;           // (B,6) dst -> src map
    float* out;                   // (B,3,oh,ow)
    int* crop_max;                // (B)
    int oh, ow;
    float inv_ow;
    float nmean[3];               // -mean
};

// the per-row / per-column fixed-point terms of WarpAffineInvoker
struct RowTerms { int X0, Y0; };

__device__ __forceinline__ RowTerms row_terms(const double* m, int y) {
#pragma clang fp contract(off)   // hipcc would fuse a*b+c into an fma; the x86 code this restates has none (an fma breaks exact .5 ties)
    const double yd = (double)y;
    const double px = m[1] * yd, py = m[4] * yd;
    // saturate_cast<int>(double) rounds half to even
    return {__double2int_rn((px + m[2]) * 1024.0) + 16, __double2int_rn((py + m[5]) * 1024.0) + 16};
}

// One output pixel in two phases so that the loads of a thread's PX pixels are all in flight together: tap_setup does the
// integer geometry and issues two 8-byte loads at an always-valid (clamped) position; tap_finish blends them, or walks the
// border path for the few pixels that need it.  (Measured on MI355X, 4096 crops of 1080p frames: ~100 VALU instructions per
// pixel occupy ~half of the 1.1 ms, waves wait on memory 44 % of their time — profiles/r01_notes.md.)
struct Tap {
    int sx, sy, fx, fy;
    bool interior;
    uint64_t q0, q1;
};

__device__ __forceinline__ void tap_setup(Tap& t, const uint8_t* __restrict__ src, RowTerms rt, int adelta, int bdelta, int sh, int sw, int mirror) {
    const int X = (rt.X0 + adelta) >> 5, Y = (rt.Y0 + bdelta) >> 5;
    t.sx = min(max(X >> 5, -32768), 32767);
    t.sy = min(max(Y >> 5, -32768), 32767);
    t.fx = X & 31; t.fy = Y & 31;
    const int xl = mirror ? sw - 2 - t.sx : t.sx;                     // left one of the two adjacent source columns
    t.interior = t.sy >= 0 && t.sy + 1 < sh && xl >= 0 && xl + 3 <= sw && sw >= 3 && sh >= 2;
    // both taps of a row are 6 adjacent bytes: one (unaligned) 8-byte load that stays inside the row.  A frame is < 2^31
    // bytes and sy < 2^15, sw < 2^24, so the pixel index fits 32 bits and a 24-bit multiply.
    const unsigned cy = (unsigned)min(max(t.sy, 0), max(sh - 2, 0)), cx = (unsigned)min(max(xl, 0), max(sw - 3, 0));
    const unsigned i0 = __umul24(cy, (unsigned)sw) + cx, i1 = i0 + (unsigned)sw;
    t.q0 = t.q1 = 0;
    if (sw >= 3 && sh >= 2) {                                         // uniform per crop
        __builtin_memcpy(&t.q0, src + (i0 + (i0 << 1)), 8);
        __builtin_memcpy(&t.q1, src + (i1 + (i1 << 1)), 8);
    }
}

// OpenCV's table weights are w00 = (32-fx)(32-fy)*32, w01 = fx(32-fy)*32, w10 = (32-fx)fy*32, w11 = fx*fy*32 (sum 2^15; the entry for a zero
// fraction is stored as 32767 / 0 / 0 / 1 because 2^15 does not fit a short) and the pixel is (sum t*w + 2^14) >> 15.  The same integer comes out
// of two nested linear interpolations with 5-bit weights:
//     h_r = 32*ta_r + fx*(tb_r - ta_r)  = ((32-fx)*ta_r + fx*tb_r)            r = upper, lower row; ta at column sx, tb at sx+1
//     v   = (32*h_0 + fy*(h_1 - h_0) + 512) >> 10
// because sum t*w = 32*((32-fy)*h_0 + fy*h_1) exactly and floor((32 A + 2^14) / 2^15) = floor((A + 512) / 1024); for the zero fraction the table
// gives (32767 t00 + t11 + 2^14) >> 15 = t00 + floor((2^14 + t11 - t00) / 2^15) = t00, which is what the nested form gives too.  Half the
// multiplies, no weight set-up.
__device__ __forceinline__ int blend(int ta0, int tb0, int ta1, int tb1, int fx, int fy) {
    const int h0 = (ta0 << 5) + __mul24(fx, tb0 - ta0), h1 = (ta1 << 5) + __mul24(fx, tb1 - ta1);      // |operands| < 2^14: 24-bit multiplies
    return ((h0 << 5) + __mul24(fy, h1 - h0) + 512) >> 10;
}

__device__ __forceinline__ void tap_finish(const Tap& t, const uint8_t* __restrict__ src, int sh, int sw, int mirror, int v[3]) {
    if (t.interior) {
        const unsigned l0 = (unsigned)t.q0, h0 = (unsigned)(t.q0 >> 32), l1 = (unsigned)t.q1, h1 = (unsigned)(t.q1 >> 32);
        // left / right loaded pixel of each row; under a mirror the LEFT one in memory is the tap at column sx + 1
        const int tl0[3] = {(int)(l0 & 255u), (int)((l0 >> 8) & 255u), (int)((l0 >> 16) & 255u)}, tr0[3] = {(int)(l0 >> 24), (int)(h0 & 255u), (int)((h0 >> 8) & 255u)};
        const int tl1[3] = {(int)(l1 & 255u), (int)((l1 >> 8) & 255u), (int)((l1 >> 16) & 255u)}, tr1[3] = {(int)(l1 >> 24), (int)(h1 & 255u), (int)((h1 >> 8) & 255u)};
#pragma unroll
        for (int c = 0; c < 3; ++c)
            v[c] = mirror ? blend(tr0[c], tl0[c], tr1[c], tl1[c], t.fx, t.fy) : blend(tl0[c], tr0[c], tl1[c], tr1[c], t.fx, t.fy);
        return;
    }
    const int sx = t.sx, sy = t.sy;
    v[0] = v[1] = v[2] = 0;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return;     // all four taps in the constant border
    const int xa = mirror ? sw - 1 - sx : sx, xb = mirror ? sw - 2 - sx : sx + 1;
    const long long r0 = (long long)sy * sw, r1 = r0 + sw;
    const bool x0ok = sx >= 0, x1ok = sx + 1 < sw, y0ok = sy >= 0, y1ok = sy + 1 < sh;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int t00 = (x0ok && y0ok) ? src[(r0 + xa) * 3 + c] : 0;
        const int t01 = (x1ok && y0ok) ? src[(r0 + xb) * 3 + c] : 0;
        const int t10 = (x0ok && y1ok) ? src[(r1 + xa) * 3 + c] : 0;
        const int t11 = (x1ok && y1ok) ? src[(r1 + xb) * 3 + c] : 0;
        v[c] = blend(t00, t01, t10, t11, t.fx, t.fy);
    }
}

// Each thread makes PX (8, or 4) consecutive pixels of one output row (out_w % PX == 0) and stores PX / 4 float4 per channel plane.
// FIXUP = false: every crop, written as v / 255 - mean, and the crop's maximum collected.
// FIXUP = true : only crops whose maximum is <= 1 (im_to_torch leaves those undivided), rewritten as v - mean.
template <bool FIXUP, int PX>
__global__ __launch_bounds__(256) void crop_warp_kernel(CropParams p) {
    extern __shared__ int lds[];                                     // [0,256): v / 255 as float bits; then adelta[ow], bdelta[ow]
    const int b = blockIdx.y;
    if (FIXUP && p.crop_max[b] > 1) return;
    const double* m = p.minv + (long long)b * 6;
    float* lut = reinterpret_cast<float*>(lds);
    int* adelta = lds + 256;
    int* bdelta = adelta + p.ow;
    lut[threadIdx.x] = FIXUP ? (float)threadIdx.x : __fdiv_rn((float)threadIdx.x, 255.0f);    // one IEEE division per thread instead of 3 per pixel
    for (int x = threadIdx.x; x < p.ow; x += 256) {                  // the per-column terms, once per block instead of per pixel
        const double xd = (double)x;
        adelta[x] = __double2int_rn(m[0] * xd * 1024.0);
        bdelta[x] = __double2int_rn(m[3] * xd * 1024.0);
    }
    __syncthreads();
    const int plane = p.oh * p.ow;
    int mx = 0;
    // (the first pass covers the crop with one step per thread; the second — which rewrites only the rare crops whose maximum is <= 1 —
    // is launched as ONE block per crop that walks it: 4096 blocks that return at once instead of 98 304)
    for (int q = (blockIdx.x * 256 + threadIdx.x) * PX; q < plane; q += gridDim.x * 256 * PX) {
        const int y = fast_div(q, p.inv_ow), x = q - y * p.ow;
        const int sh = p.src_hwf[b * 3 + 0], sw = p.src_hwf[b * 3 + 1], mirror = p.src_hwf[b * 3 + 2];
        const uint8_t* src = p.arena + p.src_off[b];
        const RowTerms rt = row_terms(m, y);
        float r[3][PX];
        Tap taps[PX];
#pragma unroll
        for (int i = 0; i < PX; ++i) tap_setup(taps[i], src, rt, adelta[x + i], bdelta[x + i], sh, sw, mirror);
#pragma unroll
        for (int i = 0; i < PX; ++i) {
            int v[3];
            tap_finish(taps[i], src, sh, sw, mirror, v);
            mx = max(mx, max(v[0], max(v[1], v[2])));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                r[c][i] = __fadd_rn(lut[v[c]], p.nmean[c]);
            }
        }
        float* o = p.out + (long long)b * 3 * plane + q;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            if (PX >= 4) {
#pragma unroll
                for (int u = 0; u < PX / 4; ++u)
                    *reinterpret_cast<f32x4*>(o + (long long)c * plane + 4 * u) = f32x4{r[c][4 * u], r[c][4 * u + 1], r[c][4 * u + 2], r[c][4 * u + 3]};
            } else o[(long long)c * plane] = r[c][0];
        }
    }
    if (!FIXUP) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
        if ((threadIdx.x & 63) == 0 && mx > 1) atomicMax(p.crop_max + b, mx);
    }
}

}  // namespace vatl

using namespace vatl;

static std::atomic<int> g_crop_px{8};          // pixels per thread of the warp kernel (vatl::crop_tune_px, profiling variant only: 4 or 8; identical results)
namespace vatl {
__attribute__((visibility("hidden"))) int crop_tune_px(int px) { if (px != 4 && px != 8) return -1; g_crop_px.store(px, std::memory_order_relaxed); return 0; }
}

extern "C" int vatl_crop_warp_affine(const uint8_t* arena, const int64_t* src_off, const int32_t* src_hwf, const double* minv, float* out,
                                     int32_t* crop_max, int B, int out_h, int out_w, float mean0, float mean1, float mean2, void* stream) {
    if (B == 0) return 0;
    if (!arena || !src_off || !src_hwf || !minv || !out || !crop_max) return fail(VATL_EINVAL, "vatl_crop_warp_affine: null pointer");
    if (B < 0 || B > 65535 || out_h <= 0 || out_w <= 0 || out_w > 4096 || (long long)out_h * out_w >= (1 << 20))
        return fail(VATL_EINVAL, "vatl_crop_warp_affine: B=%d (<= 65535), out %dx%d (width <= 4096, < 2^20 pixels)", B, out_h, out_w);
    hipStream_t s = (hipStream_t)stream;
    CropParams p{arena, (const long long*)src_off, src_hwf, minv, out, crop_max, out_h, out_w, 1.0f / (float)out_w, {-mean0, -mean1, -mean2}};
    if (hipMemsetAsync(crop_max, 0, sizeof(int) * (size_t)B, s) != hipSuccess) return fail(VATL_ELAUNCH, "vatl_crop_warp_affine: memset failed");
    const size_t lds_bytes = sizeof(int) * (256 + 2 * (size_t)out_w);
    if (out_w % 8 == 0 && ((uintptr_t)out & 15) == 0 && g_crop_px.load(std::memory_order_relaxed) == 8) {
        const dim3 grid(cdiv((long long)out_h * out_w, 2048), B);            // eight pixels per thread: half the per-thread / per-block set-up
        hipLaunchKernelGGL((crop_warp_kernel<false, 8>), grid, dim3(256), lds_bytes, s, p);
        hipLaunchKernelGGL((crop_warp_kernel<true, 8>), dim3(1, B), dim3(256), lds_bytes, s, p);
    } else if (out_w % 4 == 0 && ((uintptr_t)out & 15) == 0) {
        const dim3 grid(cdiv((long long)out_h * out_w, 1024), B);
        hipLaunchKernelGGL((crop_warp_kernel<false, 4>), grid, dim3(256), lds_bytes, s, p);
        hipLaunchKernelGGL((crop_warp_kernel<true, 4>), dim3(1, B), dim3(256), lds_bytes, s, p);
    } else {
        const dim3 grid(cdiv((long long)out_h * out_w, 256), B);
        hipLaunchKernelGGL((crop_warp_kernel<false, 1>), grid, dim3(256), lds_bytes, s, p);
        hipLaunchKernelGGL((crop_warp_kernel<true, 1>), dim3(1, B), dim3(256), lds_bytes, s, p);
    }
    return check_launch("vatl_crop_warp_affine");
}
