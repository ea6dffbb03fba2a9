// Second group of scorers / API-parity kernels:
//   tpc           compute_tpc + neighbour rule        ActiveLearning.py:333-344, 736-745
//   soft-arg-max  heatmap_to_coord_simple_regress     transforms.py:586-702  (L1JointRegression configs)
//   ae forward    WholeBodyAE.forward (+ MSE)         Whole_body_AE/AutoEncoder.py:13-39
//   hybrid f64    compute_hybrid on float64 inputs    Whole_body_AE/hybrid_feature.py:14-59
//   peak mask     localpeak_values                    local_peak.py:5-10
#include "common.h"

namespace vatl {

// --------------------------------------------------------------------------
// TPC: one thread per item.  adj_prev[i] / adj_next[i] are the neighbours' heat-maps
// decoded with item i's box (vatl_decode_argmax_affine on shifted views).
// --------------------------------------------------------------------------
__device__ __forceinline__ int moved_joints(const float* a, const float* b, int J, double thresh) {
    int c = 0;
    for (int j = 0; j < J; ++j) {
        const float dx = a[2 * j] - b[2 * j], dy = a[2 * j + 1] - b[2 * j + 1];
        const float d = sqrtf(dx * dx + dy * dy);             // np.linalg.norm on float32 rows
        c += ((double)d > thresh) ? 1 : 0;
    }
    return c;
}

__global__ void tpc_stream_kernel(const float* __restrict__ cur, const float* __restrict__ adj_prev, const float* __restrict__ adj_next,
                                  const float* __restrict__ bbox, const uint8_t* __restrict__ is_prev, const uint8_t* __restrict__ is_next,
                                  float* __restrict__ tpc, int N, int J) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double w = (double)bbox[4 * i + 2] - (double)bbox[4 * i + 0], h = (double)bbox[4 * i + 3] - (double)bbox[4 * i + 1];
    const double thresh = 0.01 * sqrt(w * h);
    const bool hp = is_prev[i] != 0 && i > 0, hn = is_next[i] != 0 && i < N - 1;
    int t = 0;
    if (hp) t += moved_joints(cur + (long long)i * J * 2, adj_prev + (long long)i * J * 2, J, thresh);
    if (hn) t += moved_joints(cur + (long long)i * J * 2, adj_next + (long long)i * J * 2, J, thresh);
    if (hp != hn) t *= 2;
    tpc[i] = (float)t;
}

// --------------------------------------------------------------------------
// soft-arg-max decode: one block per (item, joint)
// --------------------------------------------------------------------------
template <int NORM>   // 0 softmax, 1 sigmoid, 2 divide_sum
__global__ __launch_bounds__(256) void softargmax_kernel(const float* __restrict__ hm, const float* __restrict__ bbox,
                                                         float* __restrict__ coords, float* __restrict__ scores, int J, int H, int W) {
    const int item = blockIdx.x / J;
    const int HW = H * W;
    const float* gsrc = hm + (long long)blockIdx.x * HW;
    const int tid = threadIdx.x;
    extern __shared__ __attribute__((aligned(16))) float plane[];   // the plane is read from HBM once
    __shared__ float red[4];
    __shared__ double dred[3][4];
    float mx = -INFINITY;
    const int step = (HW & 3) == 0 ? 4 : 1;            // 16-byte loads when the plane allows it
    if (step == 4) {
        for (int q = tid; q < (HW >> 2); q += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(gsrc + 4 * q);
            *reinterpret_cast<f32x4*>(plane + 4 * q) = v;
            mx = fmaxf(fmaxf(mx, fmaxf(v[0], v[1])), fmaxf(v[2], v[3]));
        }
    } else {
        for (int q = tid; q < HW; q += 256) { const float v = gsrc[q]; plane[q] = v; mx = fmaxf(mx, v); }
    }
    const float* src = plane;                          // each thread re-reads only what it wrote
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    double s = 0.0, sx = 0.0, sy = 0.0;
    const float inv_w = 1.0f / (float)W;
    for (int q0 = tid * step; q0 < HW; q0 += 256 * step) {
        for (int q = q0; q < q0 + step; ++q) {
            const float v = src[q];
            float p;
            if (NORM == 0) p = expf(v - mx);
            else if (NORM == 1) p = 1.f / (1.f + expf(-v));
            else p = v;
            const int y = fast_div(q, inv_w), x = q - y * W;
            s += p; sx += (double)p * x; sy += (double)p * y;
        }
    }
    s = wave_sum(s); sx = wave_sum(sx); sy = wave_sum(sy);
    if ((tid & 63) == 0) { dred[0][tid >> 6] = s; dred[1][tid >> 6] = sx; dred[2][tid >> 6] = sy; }
    __syncthreads();
    if (tid != 0) return;
    s = dred[0][0] + dred[0][1] + dred[0][2] + dred[0][3];
    sx = dred[1][0] + dred[1][1] + dred[1][2] + dred[1][3];
    sy = dred[2][0] + dred[2][1] + dred[2][2] + dred[2][3];
    // expectation -> /W - 0.5 -> (c + 0.5) * W, in float32 like the reference's tensors
    const float ex = (float)(sx / s), ey = (float)(sy / s);
    const float u = ((ex / (float)W - 0.5f) + 0.5f) * (float)W;
    const float v = ((ey / (float)H - 0.5f) + 0.5f) * (float)H;
    const double xmin = bbox[item * 4 + 0], ymin = bbox[item * 4 + 1], xmax = bbox[item * 4 + 2], ymax = bbox[item * 4 + 3];
    const double bw = xmax - xmin, bh = ymax - ymin;
    const double cx = xmin + bw * 0.5, cy = ymin + bh * 0.5;
    const float cx32 = (float)cx, cy32 = (float)cy;
    const float top32 = (float)(cy + bw * -0.5);
    const double g = (double)(cy32 - top32) / (W * 0.5);
    coords[(long long)blockIdx.x * 2 + 0] = (float)((double)cx32 + ((double)u - W * 0.5) * g);
    coords[(long long)blockIdx.x * 2 + 1] = (float)((double)cy32 + ((double)v - H * 0.5) * g);
    scores[blockIdx.x] = NORM == 1 ? 1.f / (1.f + expf(-mx)) : 1.f;
}

// Same arithmetic with one WAVE per plane and the plane held in registers (NV float4 per lane, all loads in flight at once, no
// LDS, no block barrier): used when the plane is exactly 64 * NV float4 (64x48 -> NV = 12, 96x72 -> NV = 27).
template <int NORM, int NV>
__global__ __launch_bounds__(256) void softargmax_wave_kernel(const float* __restrict__ hm, const float* __restrict__ bbox,
                                                              float* __restrict__ coords, float* __restrict__ scores, int planes, int J, int H, int W) {
    const int lane = threadIdx.x & 63;
    const long long plane = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int item = (int)(plane / J);
    const f32x4* src = reinterpret_cast<const f32x4*>(hm + plane * (64LL * NV * 4)) + lane;
    f32x4 v[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = src[k * 64];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < NV; ++k) mx = fmaxf(fmaxf(mx, fmaxf(v[k][0], v[k][1])), fmaxf(v[k][2], v[k][3]));
    mx = wave_max(mx);
    double s = 0.0, sx = 0.0, sy = 0.0;
    const float inv_w = 1.0f / (float)W;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int q0 = 4 * (k * 64 + lane);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float val = v[k][c];
            float p;
            if (NORM == 0) p = expf(val - mx);
            else if (NORM == 1) p = 1.f / (1.f + expf(-val));
            else p = val;
            const int q = q0 + c;
            const int y = fast_div(q, inv_w), x = q - y * W;
            s += p; sx += (double)p * x; sy += (double)p * y;
        }
    }
    s = wave_sum(s); sx = wave_sum(sx); sy = wave_sum(sy);
    if (lane != 0) return;
    // expectation -> /W - 0.5 -> (c + 0.5) * W, in float32 like the reference's tensors
    const float ex = (float)(sx / s), ey = (float)(sy / s);
    const float u = ((ex / (float)W - 0.5f) + 0.5f) * (float)W;
    const float vv = ((ey / (float)H - 0.5f) + 0.5f) * (float)H;
    const double xmin = bbox[item * 4 + 0], ymin = bbox[item * 4 + 1], xmax = bbox[item * 4 + 2];
    const double bw = xmax - xmin, bh = (double)bbox[item * 4 + 3] - ymin;
    const double cx = xmin + bw * 0.5, cy = ymin + bh * 0.5;
    const float cx32 = (float)cx, cy32 = (float)cy;
    const float top32 = (float)(cy + bw * -0.5);
    const double g = (double)(cy32 - top32) / (W * 0.5);
    coords[plane * 2 + 0] = (float)((double)cx32 + ((double)u - W * 0.5) * g);
    coords[plane * 2 + 1] = (float)((double)cy32 + ((double)vv - H * 0.5) * g);
    scores[plane] = NORM == 1 ? 1.f / (1.f + expf(-mx)) : 1.f;
}

template <int NORM>
static bool launch_softargmax_wave(const float* hm, const float* bbox, float* coords, float* scores, int N, int J, int H, int W, hipStream_t st) {
    const long long planes = (long long)N * J;
    if ((((uintptr_t)hm) & 15) != 0 || planes > 0x7FFFFFFF) return false;
    if (H * W == 64 * 12 * 4) hipLaunchKernelGGL((softargmax_wave_kernel<NORM, 12>), dim3(cdiv(planes, 4)), dim3(256), 0, st, hm, bbox, coords, scores, (int)planes, J, H, W);
    else if (H * W == 64 * 27 * 4) hipLaunchKernelGGL((softargmax_wave_kernel<NORM, 27>), dim3(cdiv(planes, 4)), dim3(256), 0, st, hm, bbox, coords, scores, (int)planes, J, H, W);
    else return false;
    return true;
}

// --------------------------------------------------------------------------
// auto-encoder forward on given features: one wave per item, lane = neuron
// --------------------------------------------------------------------------
__device__ __forceinline__ float dense_lane2(const float* __restrict__ Wt, const float* __restrict__ bias, int n_out, int n_in, float h, int lane) {
    float acc = lane < n_out ? bias[lane] : 0.f;
    for (int k = 0; k < n_in; ++k) {
        const float hk = __shfl(h, k, 64);
        if (lane < n_out) acc = fmaf(Wt[lane * n_in + k], hk, acc);
    }
    return acc;
}

__global__ __launch_bounds__(256) void ae_forward_kernel(const float* __restrict__ feat, const float* __restrict__ ae, int D, int z,
                                                         float* __restrict__ recon, float* __restrict__ mse, int N) {
    const int lane = threadIdx.x & 63;
    const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= N) return;
    const float x0 = lane < D ? feat[(long long)item * D + lane] : 0.f;
    const int dims[5] = {D, 24, 12, 7, z};
    const float* w = ae;
    float h = x0;
    for (int i = 0; i < 4; ++i) {
        const int ni = dims[i], no = dims[i + 1];
        h = dense_lane2(w, w + no * ni, no, ni, h, lane);
        if (i < 3) h = fmaxf(h, 0.f);
        w += no * ni + no;
    }
    for (int i = 4; i > 0; --i) {
        const int ni = dims[i], no = dims[i - 1];
        h = dense_lane2(w, w + no * ni, no, ni, h, lane);
        h = i > 1 ? fmaxf(h, 0.f) : 1.f / (1.f + expf(-h));
        w += no * ni + no;
    }
    if (recon && lane < D) recon[(long long)item * D + lane] = h;
    const float d = lane < D ? (h - x0) * (h - x0) : 0.f;
    const float s = wave_sum(d);
    if (mse && lane == 0) mse[item] = s / (float)D;
}

// compute_hybrid on float64 key-points and an (x,y,w,h) box: one thread per item
__global__ void hybrid_f64_kernel(const double* __restrict__ kpts, const double* __restrict__ bbox_xywh, double* __restrict__ feat,
                                  int32_t* __restrict__ status, int N) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const double* kp = kpts + (long long)i * 51;
    double* f = feat + (long long)i * 42;
    const double height = bbox_xywh[4 * i + 3];
    double sw = 0.0, sx = 0.0, sy = 0.0;
    for (int j = 0; j < 17; ++j) { const double s = kp[3 * j + 2]; sw += s; sx += kp[3 * j] * s; sy += kp[3 * j + 1] * s; }
    int st = 0;
    if (!(height > 0.0)) st = 1; else if (!(sw > 0.0)) st = 2;
    if (status) status[i] = st;
    if (st) { for (int k = 0; k < 42; ++k) f[k] = __builtin_nan(""); return; }
    const double gx = sx / sw, gy = sy / sw;
    for (int j = 0; j < 17; ++j) { f[j] = (kp[3 * j] - gx) / height; f[17 + j] = (kp[3 * j + 1] - gy) / height; }
    const int tri[8][3] = {{8, 6, 12}, {6, 8, 10}, {5, 7, 9}, {7, 5, 11}, {11, 12, 14}, {12, 11, 13}, {12, 14, 16}, {11, 13, 15}};
    for (int t = 0; t < 8; ++t) {
        const double x0 = kp[3 * tri[t][0]], y0 = kp[3 * tri[t][0] + 1];
        const double x1 = kp[3 * tri[t][1]], y1 = kp[3 * tri[t][1] + 1];
        const double x2 = kp[3 * tri[t][2]], y2 = kp[3 * tri[t][2] + 1];
        const double m1 = (y1 - y0) / (x1 - x0 + 1e-6), m2 = (y2 - y1) / (x2 - x1 + 1e-6);
        f[34 + t] = atan(fabs((m1 - m2) / (1.0 + m1 * m2 + 1e-6)));
    }
}

// kept-peak mask of localpeak_values: one block per (item, joint) plane
__global__ __launch_bounds__(256) void localpeak_mask_kernel(const float* __restrict__ hm, uint8_t* __restrict__ mask, int H, int W, float order) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    __shared__ float wmax[4];
    const int tid = threadIdx.x;
    const int HW = H * W, PW = W + 2, PN = (H + 2) * PW;
    const float* src = hm + (long long)blockIdx.x * HW;
    for (int q = tid; q < PN; q += 256) {
        const int y = q / PW - 1, x = q % PW - 1;
        tile[q] = ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) ? src[y * W + x] : 0.f;
    }
    __syncthreads();
    float pmax = -INFINITY;
    for (int q = tid; q < HW; q += 256) {
        const int y = q / W, x = q - y * W;
        const float* c = tile + (y + 1) * PW + (x + 1);
        float m = fmaxf(fmaxf(c[-PW - 1], c[-PW]), c[-PW + 1]);
        m = fmaxf(m, fmaxf(c[-1], c[1]));
        m = fmaxf(m, fmaxf(fmaxf(c[PW - 1], c[PW]), c[PW + 1]));
        if (c[0] >= m) pmax = fmaxf(pmax, c[0]);
    }
    pmax = wave_max(pmax);
    if ((tid & 63) == 0) wmax[tid >> 6] = pmax;
    __syncthreads();
    pmax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
    const float thr = pmax * order;
    for (int q = tid; q < HW; q += 256) {
        const int y = q / W, x = q - y * W;
        const float* c = tile + (y + 1) * PW + (x + 1);
        float m = fmaxf(fmaxf(c[-PW - 1], c[-PW]), c[-PW + 1]);
        m = fmaxf(m, fmaxf(c[-1], c[1]));
        m = fmaxf(m, fmaxf(fmaxf(c[PW - 1], c[PW]), c[PW + 1]));
        mask[(long long)blockIdx.x * HW + q] = (pmax > -INFINITY && c[0] >= m && c[0] >= thr) ? 1 : 0;
    }
}

}  // namespace vatl

using namespace vatl;

extern "C" int vatl_tpc_stream(const float* cur, const float* adj_prev, const float* adj_next, const float* bbox,
                               const uint8_t* is_prev, const uint8_t* is_next, float* tpc, int N, int J, void* stream) {
    if (N <= 0) return 0;
    if (!cur || !bbox || !is_prev || !is_next || !tpc || (N > 1 && (!adj_prev || !adj_next))) return fail(VATL_EINVAL, "tpc_stream: null pointer");
    hipLaunchKernelGGL(tpc_stream_kernel, dim3(cdiv(N, 128)), dim3(128), 0, (hipStream_t)stream, cur, adj_prev, adj_next, bbox, is_prev, is_next, tpc, N, J);
    return check_launch("tpc_stream");
}

extern "C" int vatl_decode_softargmax(const float* hm, const float* bbox, float* coords, float* scores,
                                      int N, int J, int H, int W, int norm_type, void* stream) {
    if (N <= 0) return 0;
    if (!hm || !bbox || !coords || !scores) return fail(VATL_EINVAL, "decode_softargmax: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t smem = (size_t)H * W * sizeof(float);
    if (smem > 60 * 1024) return fail(VATL_EINVAL, "decode_softargmax: heat-map %dx%d too large for the LDS tile", H, W);
    if (norm_type == 0 && launch_softargmax_wave<0>(hm, bbox, coords, scores, N, J, H, W, st)) return check_launch("decode_softargmax");
    if (norm_type == 1 && launch_softargmax_wave<1>(hm, bbox, coords, scores, N, J, H, W, st)) return check_launch("decode_softargmax");
    if (norm_type == 2 && launch_softargmax_wave<2>(hm, bbox, coords, scores, N, J, H, W, st)) return check_launch("decode_softargmax");
    if (norm_type == 0) hipLaunchKernelGGL(softargmax_kernel<0>, dim3(N * J), dim3(256), smem, st, hm, bbox, coords, scores, J, H, W);
    else if (norm_type == 1) hipLaunchKernelGGL(softargmax_kernel<1>, dim3(N * J), dim3(256), smem, st, hm, bbox, coords, scores, J, H, W);
    else if (norm_type == 2) hipLaunchKernelGGL(softargmax_kernel<2>, dim3(N * J), dim3(256), smem, st, hm, bbox, coords, scores, J, H, W);
    else return fail(VATL_EINVAL, "decode_softargmax: norm_type must be 0 (softmax), 1 (sigmoid) or 2 (divide_sum)");
    return check_launch("decode_softargmax");
}

extern "C" int vatl_ae_forward(const float* feat, const float* ae, int D, int z, float* recon, float* mse, int N, void* stream) {
    if (N <= 0) return 0;
    if (!feat || !ae || (!recon && !mse)) return fail(VATL_EINVAL, "ae_forward: null pointer");
    if (D < 1 || D > 64 || z < 1 || z > 64) return fail(VATL_EINVAL, "ae_forward: widths must be in 1..64 (D=%d z=%d)", D, z);
    hipLaunchKernelGGL(ae_forward_kernel, dim3(cdiv(N, 4)), dim3(256), 0, (hipStream_t)stream, feat, ae, D, z, recon, mse, N);
    return check_launch("ae_forward");
}

extern "C" int vatl_hybrid_feature_f64(const double* kpts, const double* bbox_xywh, double* feat, int32_t* status, int N, void* stream) {
    if (N <= 0) return 0;
    if (!kpts || !bbox_xywh || !feat) return fail(VATL_EINVAL, "hybrid_feature_f64: null pointer");
    hipLaunchKernelGGL(hybrid_f64_kernel, dim3(cdiv(N, 64)), dim3(64), 0, (hipStream_t)stream, kpts, bbox_xywh, feat, status, N);
    return check_launch("hybrid_feature_f64");
}

extern "C" int vatl_localpeak_mask(const float* hm, uint8_t* mask, int planes, int H, int W, float order, void* stream) {
    if (planes <= 0) return 0;
    if (!hm || !mask) return fail(VATL_EINVAL, "localpeak_mask: null pointer");
    const size_t smem = (size_t)(H + 2) * (W + 2) * sizeof(float);
    if (smem > 60 * 1024) return fail(VATL_EINVAL, "localpeak_mask: map %dx%d too large for the LDS tile", H, W);
    hipLaunchKernelGGL(localpeak_mask_kernel, dim3(planes), dim3(256), smem, (hipStream_t)stream, hm, mask, H, W, order);
    return check_launch("localpeak_mask");
}
