// Fine-tune step pieces that are pure HBM streams:
//   masked MSE forward+backward   ActiveLearning.py:669  (0.5 * MSELoss(out*m, tgt*m))
//   AdamW update                  ActiveLearning.py:224-228, :673
#include "common.h"

namespace vatl {

constexpr int MSE_MAX_BLOCKS = 1024;

// grad = (o*m - t*m) * m / numel ; partial[block] = sum (o*m - t*m)^2 (double)
__global__ __launch_bounds__(256) void masked_mse_kernel(const float* __restrict__ o, const float* __restrict__ t,
                                                         const float* __restrict__ mask, float* __restrict__ grad,
                                                         double* __restrict__ partial, long long n4, int HW4, float inv_numel) {
    float acc = 0.f;
    for (long long q = blockIdx.x * 256LL + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        const float m = mask[q / HW4];                     // one mask value per (item, joint) plane
        const f32x4 a = *reinterpret_cast<const f32x4*>(o + 4 * q);
        const f32x4 b = *reinterpret_cast<const f32x4*>(t + 4 * q);
        f32x4 g;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float d = a[e] * m - b[e] * m;
            acc += d * d;
            g[e] = d * m * inv_numel;
        }
        *reinterpret_cast<f32x4*>(grad + 4 * q) = g;
    }
    const double s = wave_sum((double)acc);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(256) void mse_finish_kernel(const double* __restrict__ partial, int nblk, float* __restrict__ loss, double half_inv_numel) {
    double s = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) s += partial[i];
    s = wave_sum(s);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)((part[0] + part[1] + part[2] + part[3]) * half_inv_numel);
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long long n, float decay, float omb1, float b2, float omb2,
                                                    float bc2s, float eps, float step_size) {
    const long long n4 = n >> 2;
    for (long long q = blockIdx.x * 256LL + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        f32x4 P = *reinterpret_cast<f32x4*>(p + 4 * q);
        const f32x4 G = *reinterpret_cast<const f32x4*>(g + 4 * q);
        f32x4 M = *reinterpret_cast<f32x4*>(m + 4 * q);
        f32x4 V = *reinterpret_cast<f32x4*>(v + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            P[e] = P[e] * decay;
            M[e] = M[e] + (G[e] - M[e]) * omb1;
            V[e] = V[e] * b2 + G[e] * G[e] * omb2;
            P[e] = P[e] - step_size * (M[e] / (sqrtf(V[e]) / bc2s + eps));
        }
        *reinterpret_cast<f32x4*>(p + 4 * q) = P;
        *reinterpret_cast<f32x4*>(m + 4 * q) = M;
        *reinterpret_cast<f32x4*>(v + 4 * q) = V;
    }
    if (blockIdx.x == 0) {
        for (long long i = 4 * n4 + threadIdx.x; i < n; i += 256) {
            float P = p[i] * decay;
            const float G = g[i];
            const float M = m[i] + (G - m[i]) * omb1;
            const float V = v[i] * b2 + G * G * omb2;
            P = P - step_size * (M / (sqrtf(V) / bc2s + eps));
            p[i] = P; m[i] = M; v[i] = V;
        }
    }
}

// torch.optim.Adam (L2 decay folded into the gradient) and torch.optim.SGD (momentum, dampening 0, no Nesterov)
// ActiveLearning.py:220-223.  MODE 0 = Adam, 1 = SGD first step (buf = g), 2 = SGD later steps (buf = mu*buf + g).
template <int MODE>
__device__ __forceinline__ void opt_elem(float& P, float G, float& M, float& V, float wd, float omb1, float b2, float omb2, float bc2s,
                                         float eps, float step_size) {
    G = G + wd * P;
    if (MODE == 0) {
        M = M + (G - M) * omb1;
        V = V * b2 + G * G * omb2;
        P = P - step_size * (M / (sqrtf(V) / bc2s + eps));
    } else {
        M = (MODE == 1) ? G : M * b2 + G;          // b2 carries the momentum for SGD
        P = P - step_size * M;
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void opt_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                  float* __restrict__ v, long long n, float wd, float omb1, float b2, float omb2,
                                                  float bc2s, float eps, float step_size) {
    const long long n4 = n >> 2;
    for (long long q = blockIdx.x * 256LL + threadIdx.x; q < n4; q += (long long)gridDim.x * 256) {
        f32x4 P = *reinterpret_cast<f32x4*>(p + 4 * q);
        const f32x4 G = *reinterpret_cast<const f32x4*>(g + 4 * q);
        f32x4 M = *reinterpret_cast<f32x4*>(m + 4 * q);
        f32x4 V = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 0) V = *reinterpret_cast<f32x4*>(v + 4 * q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float pe = P[e], me = M[e], ve = V[e];
            opt_elem<MODE>(pe, G[e], me, ve, wd, omb1, b2, omb2, bc2s, eps, step_size);
            P[e] = pe; M[e] = me; V[e] = ve;
        }
        *reinterpret_cast<f32x4*>(p + 4 * q) = P;
        *reinterpret_cast<f32x4*>(m + 4 * q) = M;
        if (MODE == 0) *reinterpret_cast<f32x4*>(v + 4 * q) = V;
    }
    if (blockIdx.x == 0) {
        for (long long i = 4 * n4 + threadIdx.x; i < n; i += 256) {
            float P = p[i], M = m[i], V = (MODE == 0) ? v[i] : 0.f;
            opt_elem<MODE>(P, g[i], M, V, wd, omb1, b2, omb2, bc2s, eps, step_size);
            p[i] = P; m[i] = M;
            if (MODE == 0) v[i] = V;
        }
    }
}

// --------------------------------------------------------------------------
// L1JointRegression (alphapose/models/criterion.py:13-94): integral (soft-arg-max) regression loss, forward and
// backward in one pass, one block per (item, joint) plane.
//   q = norm(h) (softmax | sigmoid | divide_sum, transforms.py:687-702);  p = q / sum(q);  px, py = marginals;
//   c = sum_i i * p_i (IngetralCoordinate.forward);  out = c / size - 0.5;  loss = sum |out - gt| * w  (/ B)
//   backward: dL/dc = sign(out - gt) * w / size (/ B);  IngetralCoordinate.backward replaces d c / d p_i = i by
//   AMPLITUDE * (i < c ? -1 : +1) (criterion.py:31-43);  then through p = q / sum(q) and the normalisation.
// --------------------------------------------------------------------------
__device__ __forceinline__ double block_sum4(double v, double* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

template <int NORM>   // 0 softmax, 1 sigmoid, 2 divide_sum
__global__ __launch_bounds__(256) void l1_joint_regression_kernel(const float* __restrict__ hm, const float* __restrict__ gt, const float* __restrict__ vis,
                                                                  float* __restrict__ grad, float* __restrict__ pred_jts, double* __restrict__ partial,
                                                                  int J, int H, int W, float inv_b) {
    extern __shared__ float q[];                       // normalised plane (before the second normalisation)
    __shared__ double sh[4];
    __shared__ float red[4];
    const int HW = H * W, tid = threadIdx.x;
    const int b = blockIdx.x / J, j = blockIdx.x - b * J;
    const float* src = hm + (long long)blockIdx.x * HW;
    float mx = -INFINITY;
    if (NORM == 0) {
        for (int i = tid; i < HW; i += 256) mx = fmaxf(mx, src[i]);
        mx = wave_max(mx);
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    }
    double hs = 0.0;                                   // sum of the raw plane (divide_sum) / of exp (softmax)
    for (int i = tid; i < HW; i += 256) {
        const float v = src[i];
        float t;
        if (NORM == 0) t = expf(v - mx);
        else if (NORM == 1) t = 1.f / (1.f + expf(-v));
        else t = v;
        q[i] = t;
        hs += (double)t;
    }
    const double raw_sum = block_sum4(hs, sh);
    double s = 0.0;
    if (NORM == 1) s = raw_sum;                        // q = sigmoid(h): S = sum q
    else {                                             // q = t / raw_sum (softmax or divide_sum); S = sum q (== 1 up to rounding)
        const float inv = (float)(1.0 / raw_sum);
        for (int i = tid; i < HW; i += 256) { const float t = q[i] * inv; q[i] = t; s += (double)t; }
        s = block_sum4(s, sh);
    }
    // coordinates: c = sum_i i * p_i with p = q / S
    double sx = 0.0, sy = 0.0;
    for (int i = tid; i < HW; i += 256) { const int y = i / W, x = i - y * W; sx += (double)q[i] * x; sy += (double)q[i] * y; }
    sx = block_sum4(sx, sh); sy = block_sum4(sy, sh);
    const float cx = (float)(sx / s), cy = (float)(sy / s);
    const float ox = cx / (float)W - 0.5f, oy = cy / (float)H - 0.5f;
    const long long o2 = ((long long)b * J + j) * 2;
    const float gx = gt[o2], gy = gt[o2 + 1], wx = vis[o2], wy = vis[o2 + 1];
    if (tid == 0) {
        pred_jts[o2] = ox; pred_jts[o2 + 1] = oy;
        partial[blockIdx.x] = (double)(fabsf(ox - gx) * wx) + (double)(fabsf(oy - gy) * wy);
    }
    // backward
    const float sgx = (ox > gx) ? 1.f : (ox < gx ? -1.f : 0.f), sgy = (oy > gy) ? 1.f : (oy < gy ? -1.f : 0.f);
    const float gcx = sgx * wx * inv_b / (float)W * 2.f, gcy = sgy * wy * inv_b / (float)H * 2.f;   // AMPLITUDE = 2
    const float fs = (float)s;
    double pd = 0.0;                                   // sum_k p_k dP_k
    for (int i = tid; i < HW; i += 256) {
        const int y = i / W, x = i - y * W;
        const float dP = gcx * ((float)x < cx ? -1.f : 1.f) + gcy * ((float)y < cy ? -1.f : 1.f);
        pd += (double)(q[i] / fs) * dP;
    }
    pd = block_sum4(pd, sh);
    // dq_i = (dP_i - pd) / S ; then through the normalisation
    double qd = 0.0;                                   // sum_k q_k dq_k (softmax / divide_sum)
    if (NORM != 1) {
        for (int i = tid; i < HW; i += 256) {
            const int y = i / W, x = i - y * W;
            const float dP = gcx * ((float)x < cx ? -1.f : 1.f) + gcy * ((float)y < cy ? -1.f : 1.f);
            qd += (double)q[i] * (double)((dP - (float)pd) / fs);
        }
        qd = block_sum4(qd, sh);
    }
    float* dst = grad + (long long)blockIdx.x * HW;
    for (int i = tid; i < HW; i += 256) {
        const int y = i / W, x = i - y * W;
        const float dP = gcx * ((float)x < cx ? -1.f : 1.f) + gcy * ((float)y < cy ? -1.f : 1.f);
        const float dq = (dP - (float)pd) / fs;
        float dh;
        if (NORM == 0) dh = q[i] * (dq - (float)qd);
        else if (NORM == 1) dh = dq * q[i] * (1.f - q[i]);
        else dh = (dq - (float)qd) / (float)raw_sum;
        dst[i] = dh;
    }
}

__global__ __launch_bounds__(256) void l1_finish_kernel(const double* __restrict__ partial, int n, float* __restrict__ loss, double scale) {
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = wave_sum(s);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *loss = (float)((part[0] + part[1] + part[2] + part[3]) * scale);
}

// --------------------------------------------------------------------------
// WholeBodyAE fine-tune step (ActiveLearning.py:905-925: forward, MSELoss(output, input), backward, torch.optim.Adam)
// for one mini-batch, the whole step in ONE block: the 8-layer MLP has ~3 k parameters (AutoEncoder.py:13-32), they
// and the batch activations live in LDS.  Parameter layout = vatl_pack_ae order (W0,b0,...,W7,b7).
// --------------------------------------------------------------------------
constexpr int AE_MAXB = 12, AE_MAXW = 64, AE_MAXP = 2 * (64 * 24 + 24 * 12 + 12 * 7 + 7 * 64) + 2 * (24 + 12 + 7) + 64 + 64;

__global__ __launch_bounds__(256) void ae_train_step_kernel(float* __restrict__ ae, float* __restrict__ am, float* __restrict__ av,
                                                            const float* __restrict__ feat, int B, int D, int z, float step_size, float b1,
                                                            float b2, float bc2s, float eps, float* __restrict__ loss_out) {
    __shared__ float W[AE_MAXP];                     // stays at the pre-step values until every delta is formed
    __shared__ float act[9][AE_MAXB][AE_MAXW];
    __shared__ float delta[2][AE_MAXB][AE_MAXW];
    __shared__ float red[4];
    const int tid = threadIdx.x;
    const int dims[9] = {D, 24, 12, 7, z, 7, 12, 24, D};
    int offw[8], offb[8], P = 0;
    for (int l = 0; l < 8; ++l) { offw[l] = P; P += dims[l + 1] * dims[l]; offb[l] = P; P += dims[l + 1]; }
    for (int i = tid; i < P; i += 256) W[i] = ae[i];
    for (int i = tid; i < B * D; i += 256) act[0][i / D][i % D] = feat[i];
    __syncthreads();
    // forward
    for (int l = 0; l < 8; ++l) {
        const int ni = dims[l], no = dims[l + 1];
        for (int idx = tid; idx < B * no; idx += 256) {
            const int b = idx / no, o = idx - b * no;
            float acc = W[offb[l] + o];
            for (int i = 0; i < ni; ++i) acc = fmaf(W[offw[l] + o * ni + i], act[l][b][i], acc);
            if (l == 7) acc = 1.f / (1.f + expf(-acc));
            else if (l != 3) acc = fmaxf(acc, 0.f);
            act[l + 1][b][o] = acc;
        }
        __syncthreads();
    }
    // loss and output delta (through the sigmoid)
    const float inv = 1.f / (float)(B * D);
    float ls = 0.f;
    for (int idx = tid; idx < B * D; idx += 256) {
        const int b = idx / D, o = idx - b * D;
        const float y = act[8][b][o], d = y - act[0][b][o];
        ls += d * d;
        delta[0][b][o] = 2.f * d * inv * y * (1.f - y);
    }
    ls = wave_sum(ls);
    if ((tid & 63) == 0) red[tid >> 6] = ls;
    __syncthreads();
    if (tid == 0 && loss_out) *loss_out = (red[0] + red[1] + red[2] + red[3]) * inv;
    // backward
    int cur = 0;
    for (int l = 7; l >= 0; --l) {
        const int ni = dims[l], no = dims[l + 1];
        for (int idx = tid; idx < no * ni + no; idx += 256) {
            float g = 0.f;
            int pi;
            if (idx < no * ni) { const int o = idx / ni, i = idx - o * ni; for (int b = 0; b < B; ++b) g = fmaf(delta[cur][b][o], act[l][b][i], g); pi = offw[l] + idx; }
            else { const int o = idx - no * ni; for (int b = 0; b < B; ++b) g += delta[cur][b][o]; pi = offb[l] + o; }
            // torch.optim.Adam (no weight decay); the updated value goes to global memory, W keeps the old one
            const float m = am[pi] + (g - am[pi]) * (1.f - b1);
            const float v = av[pi] * b2 + g * g * (1.f - b2);
            am[pi] = m; av[pi] = v;
            ae[pi] = W[pi] - step_size * (m / (sqrtf(v) / bc2s + eps));
        }
        if (l > 0) {
            for (int idx = tid; idx < B * ni; idx += 256) {
                const int b = idx / ni, i = idx - b * ni;
                float d = 0.f;
                for (int o = 0; o < no; ++o) d = fmaf(W[offw[l] + o * ni + i], delta[cur][b][o], d);
                if (l != 4) d = act[l][b][i] > 0.f ? d : 0.f;              // act[l] is the ReLU output of layer l-1 (act[4] = z: linear)
                delta[cur ^ 1][b][i] = d;
            }
        }
        __syncthreads();
        cur ^= 1;
    }
}

// --------------------------------------------------------------------------
// Gaussian heat-map targets (simple_transform.py:122-158): one block per (person, joint) plane
// --------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gaussian_target_kernel(const float* __restrict__ joints, const float* __restrict__ vis, float* __restrict__ target,
                                                              float* __restrict__ weight, int H, int W, double stride_x, double stride_y, float sigma) {
    const float jx = joints[2 * (long long)blockIdx.x], jy = joints[2 * (long long)blockIdx.x + 1];
    const int mu_x = (int)((double)jx / stride_x + 0.5), mu_y = (int)((double)jy / stride_y + 0.5);     // int(): truncation toward zero
    const double tmp = (double)sigma * 3.0;
    const int ulx = (int)(mu_x - tmp), uly = (int)(mu_y - tmp), brx = (int)(mu_x + tmp + 1), bry = (int)(mu_y + tmp + 1);
    float w = vis[blockIdx.x];
    const bool outside = ulx >= W || uly >= H || brx < 0 || bry < 0;
    if (outside) w = 0.f;
    if (threadIdx.x == 0) weight[blockIdx.x] = w;
    const bool draw = !outside && w > 0.5f;
    const int half = (int)(2 * tmp + 1) / 2;                      // size // 2 with size = 2*tmp + 1 (a float in the reference)
    const float inv = 1.f / (2.f * sigma * sigma);
    float* dst = target + (long long)blockIdx.x * H * W;
    const float inv_w = 1.0f / (float)W;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const int y = fast_div(i, inv_w), x = i - y * W;
        float v = 0.f;
        if (draw && x >= ulx && x < brx && y >= uly && y < bry) {
            const float dx = (float)(x - ulx - half), dy = (float)(y - uly - half);
            v = expf(-(dx * dx + dy * dy) * inv);
        }
        dst[i] = v;
    }
}

// Multi-tensor AdamW: one launch per parameter group instead of one per tensor (161 tensors for SimplePose-R50).
// table[t] = {p, g, m, v, n, first_block} (device pointers, element count and the running sum of the preceding tensors' block
// counts, all int64); a block updates kAdamBlock consecutive elements of ONE tensor, found by bisection over first_block — so a
// 8.4 M-element deconv weight and a 64-element BatchNorm bias in the same group both get blocks in proportion to their size
// (the first version gave every tensor the same <= 64 blocks: the 10.5 M-parameter deconv group ran at 1.9 TB/s on 64 CUs).
constexpr int kAdamBlock = 8192;                  // elements per block: 256 threads x 8 float4
__global__ __launch_bounds__(256) void adamw_multi_kernel(const long long* __restrict__ table, int n_tensors, float decay, float omb1, float b2,
                                                          float omb2, float bc2s, float eps, float step_size) {
    __shared__ int st;
    if (threadIdx.x == 0) {
        int lo = 0, hi = n_tensors - 1;
        const long long b = blockIdx.x;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (table[6 * (long long)mid + 5] <= b) lo = mid; else hi = mid - 1;
        }
        st = lo;
    }
    __syncthreads();
    const long long* row = table + 6 * (long long)st;
    float* __restrict__ p = reinterpret_cast<float*>(row[0]);
    const float* __restrict__ g = reinterpret_cast<const float*>(row[1]);
    float* __restrict__ m = reinterpret_cast<float*>(row[2]);
    float* __restrict__ v = reinterpret_cast<float*>(row[3]);
    const long long n = row[4];
    const long long e0 = ((long long)blockIdx.x - row[5]) * kAdamBlock;
    const long long e1 = e0 + kAdamBlock < n ? e0 + kAdamBlock : n;
    const bool vec = ((row[0] | row[1] | row[2] | row[3]) & 15) == 0;          // e0 is a multiple of 4
    long long done = e0;
    if (vec) {
        const long long q1 = e1 >> 2;
        for (long long q = (e0 >> 2) + threadIdx.x; q < q1; q += 256) {
            f32x4 P = *reinterpret_cast<f32x4*>(p + 4 * q);
            const f32x4 G = *reinterpret_cast<const f32x4*>(g + 4 * q);
            f32x4 M = *reinterpret_cast<f32x4*>(m + 4 * q);
            f32x4 V = *reinterpret_cast<f32x4*>(v + 4 * q);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                P[e] = P[e] * decay;
                M[e] = M[e] + (G[e] - M[e]) * omb1;
                V[e] = V[e] * b2 + G[e] * G[e] * omb2;
                P[e] = P[e] - step_size * (M[e] / (sqrtf(V[e]) / bc2s + eps));
            }
            *reinterpret_cast<f32x4*>(p + 4 * q) = P;
            *reinterpret_cast<f32x4*>(m + 4 * q) = M;
            *reinterpret_cast<f32x4*>(v + 4 * q) = V;
        }
        done = q1 << 2;
    }
    for (long long i = done + threadIdx.x; i < e1; i += 256) {
        float P = p[i] * decay;
        const float G = g[i];
        const float M = m[i] + (G - m[i]) * omb1;
        const float V = v[i] * b2 + G * G * omb2;
        P = P - step_size * (M / (sqrtf(V) / bc2s + eps));
        p[i] = P; m[i] = M; v[i] = V;
    }
}

}  // namespace vatl

using namespace vatl;

extern "C" int64_t vatl_masked_mse_workspace_floats(int64_t numel) { (void)numel; return 2 * MSE_MAX_BLOCKS; }

extern "C" int vatl_masked_mse_fwd_bwd(const float* out, const float* target, const float* mask, float* grad, float* loss,
                                       float* partial, int N, int J, int HW, void* stream) {
    if (!out || !target || !mask || !grad || !loss || !partial) return fail(VATL_EINVAL, "masked_mse_fwd_bwd: null pointer");
    if (HW & 3) return fail(VATL_EINVAL, "masked_mse_fwd_bwd: H*W %d must be a multiple of 4", HW);
    if ((uintptr_t)partial & 7) return fail(VATL_EINVAL, "masked_mse_fwd_bwd: workspace must be 8-byte aligned");
    const long long numel = (long long)N * J * HW;
    if (numel <= 0) return fail(VATL_EINVAL, "masked_mse_fwd_bwd: empty batch");
    const long long n4 = numel / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > MSE_MAX_BLOCKS) blocks = MSE_MAX_BLOCKS;
    hipLaunchKernelGGL(masked_mse_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, target, mask, grad,
                       reinterpret_cast<double*>(partial), n4, HW / 4, (float)(1.0 / (double)numel));
    hipLaunchKernelGGL(mse_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const double*>(partial), blocks, loss,
                       0.5 / (double)numel);
    return check_launch("masked_mse_fwd_bwd");
}

extern "C" int vatl_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                               double eps, double weight_decay, int step, void* stream) {
    if (!p || !g || !m || !v) return fail(VATL_EINVAL, "adamw_step: null pointer");
    if (step < 1) return fail(VATL_EINVAL, "adamw_step: step is 1-based");
    if (n <= 0) return 0;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return fail(VATL_EINVAL, "adamw_step: spans must be 16-byte aligned");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n,
                       (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2),
                       (float)sqrt(bc2), (float)eps, (float)(lr / bc1));
    return check_launch("adamw_step");
}

static unsigned opt_blocks(int64_t n) {
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    return (unsigned)(blocks < 1 ? 1 : blocks);
}

extern "C" int vatl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                              double eps, double weight_decay, int step, void* stream) {
    if (!p || !g || !m || !v) return fail(VATL_EINVAL, "adam_step: null pointer");
    if (step < 1) return fail(VATL_EINVAL, "adam_step: step is 1-based");
    if (n <= 0) return 0;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) return fail(VATL_EINVAL, "adam_step: spans must be 16-byte aligned");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    hipLaunchKernelGGL(opt_kernel<0>, dim3(opt_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, (float)weight_decay,
                       (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (float)(lr / bc1));
    return check_launch("adam_step");
}

extern "C" int vatl_sgd_step(float* p, const float* g, float* buf, int64_t n, double lr, double momentum, double weight_decay, int step,
                             void* stream) {
    if (!p || !g || !buf) return fail(VATL_EINVAL, "sgd_step: null pointer");
    if (step < 1) return fail(VATL_EINVAL, "sgd_step: step is 1-based");
    if (n <= 0) return 0;
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) & 15) return fail(VATL_EINVAL, "sgd_step: spans must be 16-byte aligned");
    if (step == 1)
        hipLaunchKernelGGL(opt_kernel<1>, dim3(opt_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, (float*)nullptr, (long long)n,
                           (float)weight_decay, 0.f, (float)momentum, 0.f, 1.f, 0.f, (float)lr);
    else
        hipLaunchKernelGGL(opt_kernel<2>, dim3(opt_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, buf, (float*)nullptr, (long long)n,
                           (float)weight_decay, 0.f, (float)momentum, 0.f, 1.f, 0.f, (float)lr);
    return check_launch("sgd_step");
}

extern "C" int vatl_l1_joint_regression_fwd_bwd(const float* hm, const float* gt_joints, const float* gt_joints_vis, float* grad, float* loss,
                                                float* pred_jts, double* partial, int B, int J, int H, int W, int norm_type, int size_average,
                                                void* stream) {
    if (!hm || !gt_joints || !gt_joints_vis || !grad || !loss || !pred_jts || !partial) return fail(VATL_EINVAL, "l1_joint_regression: null pointer");
    if (B <= 0 || J <= 0) return fail(VATL_EINVAL, "l1_joint_regression: empty batch");
    const size_t smem = (size_t)H * W * sizeof(float);
    if (smem > 60 * 1024) return fail(VATL_EINVAL, "l1_joint_regression: heat-map %dx%d too large for the LDS tile", H, W);
    hipStream_t st = (hipStream_t)stream;
    const float inv_b = size_average ? 1.f / (float)B : 1.f;
    const dim3 grid((unsigned)(B * J));
    if (norm_type == 0) hipLaunchKernelGGL(l1_joint_regression_kernel<0>, grid, dim3(256), smem, st, hm, gt_joints, gt_joints_vis, grad, pred_jts, partial, J, H, W, inv_b);
    else if (norm_type == 1) hipLaunchKernelGGL(l1_joint_regression_kernel<1>, grid, dim3(256), smem, st, hm, gt_joints, gt_joints_vis, grad, pred_jts, partial, J, H, W, inv_b);
    else if (norm_type == 2) hipLaunchKernelGGL(l1_joint_regression_kernel<2>, grid, dim3(256), smem, st, hm, gt_joints, gt_joints_vis, grad, pred_jts, partial, J, H, W, inv_b);
    else return fail(VATL_EINVAL, "l1_joint_regression: norm_type must be 0 (softmax), 1 (sigmoid) or 2 (divide_sum)");
    hipLaunchKernelGGL(l1_finish_kernel, dim3(1), dim3(256), 0, st, partial, B * J, loss, size_average ? 1.0 / (double)B : 1.0);
    return check_launch("l1_joint_regression");
}

extern "C" int vatl_ae_train_step(float* ae, float* m, float* v, const float* feat, int B, int D, int z, double lr, double beta1, double beta2,
                                  double eps, int step, float* loss_or_null, void* stream) {
    if (!ae || !m || !v || !feat) return fail(VATL_EINVAL, "ae_train_step: null pointer");
    if (B < 1 || B > AE_MAXB) return fail(VATL_EINVAL, "ae_train_step: batch %d must be in 1..%d (the reference trains with 10)", B, AE_MAXB);
    if (D < 1 || D > AE_MAXW || z < 1 || z > AE_MAXW) return fail(VATL_EINVAL, "ae_train_step: widths must be in 1..64 (D=%d z=%d)", D, z);
    if (step < 1) return fail(VATL_EINVAL, "ae_train_step: step is 1-based");
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    hipLaunchKernelGGL(ae_train_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, ae, m, v, feat, B, D, z, (float)(lr / bc1), (float)beta1,
                       (float)beta2, (float)sqrt(bc2), (float)eps, loss_or_null);
    return check_launch("ae_train_step");
}

extern "C" int vatl_gaussian_targets(const float* joints_xy, const float* vis, float* target, float* weight, int N, int J, int H, int W,
                                     int in_h, int in_w, float sigma, void* stream) {
    if (N <= 0) return 0;
    if (!joints_xy || !vis || !target || !weight || J <= 0 || sigma <= 0.f) return fail(VATL_EINVAL, "gaussian_targets: bad arguments");
    hipLaunchKernelGGL(gaussian_target_kernel, dim3((unsigned)(N * J)), dim3(256), 0, (hipStream_t)stream, joints_xy, vis, target, weight, H, W,
                       (double)in_h / (double)H, (double)in_w / (double)W, sigma);   // the reference divides x by stride[0] = in_h / H (:130)
    return check_launch("gaussian_targets");
}

extern "C" int vatl_adamw_step_multi(const int64_t* table_dev, int n_tensors, int64_t total_blocks, double lr, double beta1, double beta2, double eps,
                                     double weight_decay, int step, void* stream) {
    if (n_tensors <= 0) return 0;
    if (!table_dev) return fail(VATL_EINVAL, "adamw_step_multi: null table");
    if (step < 1) return fail(VATL_EINVAL, "adamw_step_multi: step is 1-based");
    if (total_blocks <= 0 || total_blocks > 0x7FFFFFFF) return fail(VATL_EINVAL, "adamw_step_multi: total_blocks %lld out of range", (long long)total_blocks);
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    hipLaunchKernelGGL(adamw_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(table_dev), n_tensors, (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps, (float)(lr / bc1));
    return check_launch("adamw_step_multi");
}

extern "C" int64_t vatl_adamw_multi_block_elems(void) { return kAdamBlock; }
