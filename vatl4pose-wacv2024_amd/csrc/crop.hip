// Crop producer — the step right before the backbone:
//   SimpleTransform.test_transform / __call__     alphapose/utils/presets/simple_transform.py:81-98, 179-251
//     cv2.warpAffine(img, trans, (w, h), flags=cv2.INTER_LINEAR)   -> fixed-point bilinear, BORDER_CONSTANT 0
//     im_to_torch (alphapose/utils/transforms.py:76-91)            -> HWC u8 -> CHW fp32, / 255 when the crop's max > 1
//     img[c].add_(-mean[c])                                        -> simple_transform.py:93-95
// The warp restates OpenCV 4.8's integer algorithm (imgwarp.cpp WarpAffineInvoker + remapBilinear<FixedPtCast<int,uchar,15>>):
// 10-bit fixed-point source coordinates rounded to 1/32 px, int16 weights scaled by 2^15.  All of it is integer work
// after the two double multiplies per row / column, so the u8 result is defined bit for bit.
// HBM-bound and write-dominated: 12 B per output pixel against <= 12 source bytes that mostly hit in L2.
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

// one output pixel: the three u8 channel values of cv2.warpAffine
__device__ __forceinline__ void warp_pixel(const CropParams& p, int b, int x, int y, int v[3]) {
#pragma clang fp contract(off)   // hipcc would fuse a*b+c into an fma; the x86 code this restates has none (an fma breaks exact .5 ties)
    const double* m = p.minv + (long long)b * 6;
    const int sh = p.src_hwf[b * 3 + 0], sw = p.src_hwf[b * 3 + 1], mirror = p.src_hwf[b * 3 + 2];
    const double xd = (double)x, yd = (double)y;
    // saturate_cast<int>(double) rounds half to even
    const int adelta = __double2int_rn(m[0] * xd * 1024.0);
    const int bdelta = __double2int_rn(m[3] * xd * 1024.0);
    const double px = m[1] * yd, py = m[4] * yd;
    const int X0 = __double2int_rn((px + m[2]) * 1024.0) + 16;
    const int Y0 = __double2int_rn((py + m[5]) * 1024.0) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
    if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }                    // the table entry for a zero fraction (short saturation)
    v[0] = v[1] = v[2] = 0;
    if (sx >= sw || sx + 1 < 0 || sy >= sh || sy + 1 < 0) return;     // all four taps in the constant border
    const uint8_t* src = p.arena + p.src_off[b];
    const bool x0ok = sx >= 0, x1ok = sx + 1 < sw, y0ok = sy >= 0, y1ok = sy + 1 < sh;
    const int xa = mirror ? sw - 1 - sx : sx, xb = mirror ? sw - 2 - sx : sx + 1;
    const long long r0 = (long long)sy * sw, r1 = r0 + sw;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int t00 = (x0ok && y0ok) ? src[(r0 + xa) * 3 + c] : 0;
        const int t01 = (x1ok && y0ok) ? src[(r0 + xb) * 3 + c] : 0;
        const int t10 = (x0ok && y1ok) ? src[(r1 + xa) * 3 + c] : 0;
        const int t11 = (x1ok && y1ok) ? src[(r1 + xb) * 3 + c] : 0;
        const int acc = t00 * w00 + t01 * w01 + t10 * w10 + t11 * w11;
        v[c] = min(max((acc + (1 << 14)) >> 15, 0), 255);
    }
}

// FIXUP = false: every crop, written as v / 255 - mean, and the crop's maximum collected.
// FIXUP = true : only crops whose maximum is <= 1 (im_to_torch leaves those undivided), rewritten as v - mean.
template <bool FIXUP>
__global__ __launch_bounds__(256) void crop_warp_kernel(CropParams p) {
    const int b = blockIdx.y;
    if (FIXUP && p.crop_max[b] > 1) return;
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int plane = p.oh * p.ow;
    int v[3] = {0, 0, 0};
    if (q < plane) {
        const int y = fast_div(q, p.inv_ow), x = q - y * p.ow;
        warp_pixel(p, b, x, y, v);
        float* o = p.out + (long long)b * 3 * plane + q;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float f = FIXUP ? (float)v[c] : __fdiv_rn((float)v[c], 255.0f);
            o[(long long)c * plane] = __fadd_rn(f, p.nmean[c]);
        }
    }
    if (!FIXUP) {
        int mx = max(v[0], max(v[1], v[2]));
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
    const dim3 grid(cdiv((long long)out_h * out_w, 256), B);
    hipLaunchKernelGGL(crop_warp_kernel<false>, grid, dim3(256), 0, s, p);
    hipLaunchKernelGGL(crop_warp_kernel<true>, grid, dim3(256), 0, s, p);
    return check_launch("vatl_crop_warp_affine");
}
