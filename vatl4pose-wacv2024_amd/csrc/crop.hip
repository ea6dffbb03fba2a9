// Crop producer — the step right before the backbone:
//   SimpleTransform.test_transform / __call__     alphapose/utils/presets/simple_transform.py:81-98, 179-251
//     cv2.warpAffine(img, trans, (w, h), flags=cv2.INTER_LINEAR)   -> fixed-point bilinear, BORDER_CONSTANT 0
//     im_to_torch (alphapose/utils/transforms.py:76-91)            -> HWC u8 -> CHW fp32, / 255 when the crop's max > 1
//     img[c].add_(-mean[c])                                        -> simple_transform.py:93-95
// The warp restates OpenCV 4.8's integer algorithm (imgwarp.cpp WarpAffineInvoker + remapBilinear<FixedPtCast<int,uchar,15>>):
// 10-bit fixed-point source coordinates rounded to 1/32 px, int16 weights scaled by 2^15.  All of it is integer work
// after the two double multiplies per row / column, so the u8 result is defined bit for bit.
// Write-dominated: 12 B per output pixel against <= 12 source bytes that mostly hit in L2; 2.7 TB/s measured (integer-ALU and
// latency mix, not yet the HBM roof).
#include "common.h"

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
    int sx, sy, w00, w01, w10, w11;
    bool interior;
    uint64_t q0, q1;
};

__device__ __forceinline__ void tap_setup(Tap& t, const uint8_t* __restrict__ src, RowTerms rt, int adelta, int bdelta, int sh, int sw, int mirror) {
    const int X = (rt.X0 + adelta) >> 5, Y = (rt.Y0 + bdelta) >> 5;
    t.sx = min(max(X >> 5, -32768), 32767);
    t.sy = min(max(Y >> 5, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    // (32-fx)(32-fy)*32, fx(32-fy)*32, (32-fx)fy*32, fx*fy*32 from one 24-bit multiply
    const int fxy = __mul24(fx, fy);
    t.w11 = fxy << 5; t.w01 = ((fx << 5) - fxy) << 5; t.w10 = ((fy << 5) - fxy) << 5; t.w00 = (1024 - ((fx + fy) << 5) + fxy) << 5;
    if ((fx | fy) == 0) { t.w00 = 32767; t.w11 = 1; }                // the table entry for a zero fraction (short saturation)
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

__device__ __forceinline__ void tap_finish(const Tap& t, const uint8_t* __restrict__ src, int sh, int sw, int mirror, int v[3]) {
    if (t.interior) {
        const unsigned l0 = (unsigned)t.q0, h0 = (unsigned)(t.q0 >> 32), l1 = (unsigned)t.q1, h1 = (unsigned)(t.q1 >> 32);
        const unsigned wl0 = mirror ? t.w01 : t.w00, wr0 = mirror ? t.w00 : t.w01, wl1 = mirror ? t.w11 : t.w10, wr1 = mirror ? t.w10 : t.w11;
        const unsigned tl0[3] = {l0 & 255u, (l0 >> 8) & 255u, (l0 >> 16) & 255u}, tr0[3] = {l0 >> 24, h0 & 255u, (h0 >> 8) & 255u};
        const unsigned tl1[3] = {l1 & 255u, (l1 >> 8) & 255u, (l1 >> 16) & 255u}, tr1[3] = {l1 >> 24, h1 & 255u, (h1 >> 8) & 255u};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const unsigned acc = __umul24(tl0[c], wl0) + __umul24(tr0[c], wr0) + __umul24(tl1[c], wl1) + __umul24(tr1[c], wr1);
            v[c] = (int)((acc + (1u << 14)) >> 15);                   // weights are >= 0 and sum to 2^15: already within 0..255
        }
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
        const int acc = t00 * t.w00 + t01 * t.w01 + t10 * t.w10 + t11 * t.w11;
        v[c] = min(max((acc + (1 << 14)) >> 15, 0), 255);
    }
}

// Each thread makes 4 consecutive pixels of one output row (out_w % 4 == 0) and stores one float4 per channel plane.
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
    const int q = (blockIdx.x * 256 + threadIdx.x) * PX;
    const int plane = p.oh * p.ow;
    int mx = 0;
    if (q < plane) {
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
            if (PX == 4) *reinterpret_cast<f32x4*>(o + (long long)c * plane) = f32x4{r[c][0], r[c][1], r[c][2], r[c][3]};
            else o[(long long)c * plane] = r[c][0];
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
    if (out_w % 4 == 0 && ((uintptr_t)out & 15) == 0) {
        const dim3 grid(cdiv((long long)out_h * out_w, 1024), B);
        hipLaunchKernelGGL((crop_warp_kernel<false, 4>), grid, dim3(256), lds_bytes, s, p);
        hipLaunchKernelGGL((crop_warp_kernel<true, 4>), grid, dim3(256), lds_bytes, s, p);
    } else {
        const dim3 grid(cdiv((long long)out_h * out_w, 256), B);
        hipLaunchKernelGGL((crop_warp_kernel<false, 1>), grid, dim3(256), lds_bytes, s, p);
        hipLaunchKernelGGL((crop_warp_kernel<true, 1>), grid, dim3(256), lds_bytes, s, p);
    }
    return check_launch("vatl_crop_warp_affine");
}
