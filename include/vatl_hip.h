/*
 * vatl_hip.h — C ABI of libvatl_hip.so, the MI355X (gfx950) implementation of
 * the VATL4Pose pose-inference + uncertainty-scoring hot path.
 *
 * The reference (ImIntheMiddle/VATL4Pose-WACV2024) is pure Python; it has no FFI
 * of its own for this path.  Each entry point below therefore names the
 * reference Python call site whose arithmetic it replaces (paths relative to the
 * reference root); INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocator);
 *     the library never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - return 0 on success, a negative VATL_E* code on failure; the message of
 *     the last failure on the calling thread is vatl_last_error();
 *   - activations between conv entry points are NHWC fp32 ("channels-last"),
 *     the network input and the heat-maps are NCHW fp32 like the reference's;
 *   - re-entrant.  Mutable state: the thread-local error string, FLOP meter and split-K / stream-K workspace registrations of
 *     the calling host thread; and two PROCESS-GLOBAL pieces — the route selectors of vatl_tune_set (every accepted value
 *     computes the same bits) and the opt-in device-wide split-K workspace of vatl_set_splitk_workspace (changes bits; off
 *     unless the caller registers one).  Nothing in the library reads the environment.
 */
#ifndef VATL_HIP_H
#define VATL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VATL_VERSION 100          /* 0.1.0 */
#define VATL_EINVAL  (-1)         /* bad argument / unsupported shape */
#define VATL_ELAUNCH (-2)         /* hipLaunch / hipFuncSetAttribute failed */

int         vatl_version(void);
const char* vatl_last_error(void);

/* Executed-FLOP meter of the CALLING HOST THREAD (thread-local, off by default): between _begin and _end every
 * matrix-core launch this thread makes adds the multiply-adds its grid really executes on the MFMA pipe, tile
 * padding included (2 * padded rows * padded columns * padded reduction), split into direct-sum launches
 * (implicit GEMM, weight gradients) and Winograd transform-domain launches.  bench.py divides these by the
 * measured time: the roofline fraction of the fp32 matrix pipe (the reference has no counterpart; its convs are
 * cuDNN calls behind nn.Conv2d, Resnet.py:63-100).  Any of the four outputs may be NULL. */
int vatl_flop_meter_begin(void);
int vatl_flop_meter_end(double* direct_flops, double* winograd_flops, int64_t* direct_launches, int64_t* winograd_launches);
/* Launch counts per kernel family since the calling thread's last _begin (read BEFORE _end): counts[k] for route k of
 * VATL_ROUTE_NAMES below, k < n; returns the number of routes the library knows.  Diagnostic: lets a test assert that the
 * configuration it pins really took the route it names (two-half Winograd blocks, staging-address tables, the
 * BatchNorm-backward epilogue, ...) instead of trusting the dispatch rules. */
#define VATL_ROUTE_NAMES "igemm,igemm_bnbwd,igemm_dma,persistent_1x1,streamk,rows_1x1,bottleneck_chain,stem_pool,halo_3x3," \
                         "winograd,winograd_2h,winograd_bnbwd,winograd_persist,winograd_c32,wgrad,winograd_wgrad,winograd_wgrad_2h," \
                         "winograd_wgrad_table,winograd_f4,winograd_f4_bnbwd"
int vatl_flop_meter_routes(int64_t* counts, int n);

/* ------------------------------------------------------------------------ *
 * Layout and parameter preparation (done once per weight version)
 * ------------------------------------------------------------------------ */

/* (N,C,H,W) -> (N,H,W,Cpad), channels C..Cpad-1 zero.  Network input:
 * ActiveLearning.py:277 `m(inps[:,0].cuda())` hands over NCHW crops. */
int vatl_nchw_to_nhwc(const float* src, float* dst, int N, int C, int H, int W, int Cpad, void* stream);
/* (N,H,W,C) -> (N,C,H,W). */
int vatl_nhwc_to_nchw(const float* src, float* dst, int N, int C, int H, int W, void* stream);

/* nn.Conv2d weight (Cout,Cin,R,S) -> [CoutPad][R][Spad][CinPad] (zeros in the
 * padding), the K-contiguous layout the implicit-GEMM kernel reads.
 * Resnet.py:63,72,97 / simplepose.py:33 weights. */
int vatl_pack_conv_weight(const float* w_oihw, float* w_packed, int Cout, int Cin, int R, int S,
                          int CoutPad, int Spad, int CinPad, void* stream);
/* nn.ConvTranspose2d(k=4,s=2,p=1) weight (Cin,Cout,4,4) -> four sub-pixel 2x2
 * filters [phase=py*2+px][CoutPad][ty][tx][Cin] with ky = 3-py-2*ty, kx = 3-px-2*tx.
 * simplepose.py:39-46. */
int vatl_pack_deconv4x4s2_weight(const float* w_iohw, float* w_packed, int Cin, int Cout, int CoutPad, void* stream);
/* The ResNet stem for inference in one launch: NCHW crops -> conv 7x7 / stride 2 / pad 3 (3 -> 64, bias-free) -> folded BatchNorm ->
 * ReLU -> max-pool 3x3 / stride 2 / pad 1 -> NHWC (N, H/4, W/4, 64)  (Resnet.py:155-158, 171-172: conv1, bn1, relu, maxpool).  The
 * filter is packed once by vatl_pack_stem_pool_weight from the (64,3,7,7) OIHW weight into vatl_stem_pool_weight_floats() floats;
 * scale / bias = vatl_bn_fold of bn1.  Served sizes: vatl_stem_pool_supported(H, W) != 0 (W / 2 in {32, 64, 96}, H % 4 == 0; 256x192
 * crops: yes, 384x288: no — callers keep vatl_nchw_to_nhwc + vatl_conv2d_fwd + vatl_maxpool3x3s2_fwd there).  x must be 8-byte aligned. */
int64_t vatl_stem_pool_weight_floats(void);
int vatl_pack_stem_pool_weight(const float* w_oihw, float* packed, void* stream);
int vatl_stem_pool_supported(int H, int W);
int vatl_stem7x7s2_pool_fwd(const float* x_nchw, const float* w_packed, const float* scale, const float* bias, float* y_nhwc,
                            int N, int H, int W, void* stream);
/* HRNet's first stem layer the same way: NCHW crops -> conv 3x3 / stride 2 / pad 1 (3 -> 64) -> folded BatchNorm -> ReLU -> NHWC
 * (N, H/2, W/2, 64)  (hrnet.py:109-110, 426-428: conv1, bn1, relu).  Filter (64,3,3,3) packed by vatl_pack_stem3_weight into
 * vatl_stem3_weight_floats() floats; same served sizes (vatl_stem_pool_supported). */
int64_t vatl_stem3_weight_floats(void);
int vatl_pack_stem3_weight(const float* w_oihw, float* packed, void* stream);
int vatl_stem3x3s2_fwd(const float* x_nchw, const float* w_packed, const float* scale, const float* bias, float* y_nhwc,
                       int N, int H, int W, void* stream);

/* Eval-mode BatchNorm2d as a per-channel affine: scale = gamma/sqrt(var+eps),
 * bias = beta - mean*scale (Resnet.py:67,98,100,155; simplepose.py:41,44,47).
 * gamma/beta may be NULL (then 1 / 0); conv_bias (may be NULL) is folded in:
 * bias += conv_bias*scale. */
int vatl_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var, const float* conv_bias,
                 float eps, float* scale, float* bias, int C, void* stream);

/* ------------------------------------------------------------------------ *
 * Backbone forward (inference): implicit-GEMM on fp32 MFMA
 * ------------------------------------------------------------------------ */

/* y = act( conv(x, w) * scale + bias (+ residual) ).
 * x (N,H,W,Cin) NHWC; w packed by vatl_pack_conv_weight with CinPad == Cin
 * (Cin % 32 == 0) or, for the 3-channel stem, Cin = 4 / Spad = 8;
 * y NHWC (N,Ho,Wo,Cout) or, when out_nchw != 0, NCHW (N,Cout,Ho,Wo);
 * scale/bias (Cout) may be NULL (1 / 0); residual (same layout as y) may be NULL.
 * Replaces Conv2d+BatchNorm2d(+add)(+ReLU) chains of Bottleneck.forward
 * (Resnet.py:104-128), ResNet.forward stem (Resnet.py:171-172) and
 * SimplePose.final_layer (simplepose.py:85). */
/* Last 1x1 conv of a bottleneck fused with the block's projection shortcut (Resnet.py:104-128 with `downsample`,
 * :185-189): y = act(W1' a + W2' x[::stride2, ::stride2] + bias), a (N,Ho,Wo,C1), x (N,H2,W2,C2), ONE implicit GEMM over
 * K = C1 + C2 with the A operand gathered from both tensors, so the projection output never exists in HBM.
 * vatl_pack_conv1x1_dual_weight builds w [CoutPad][C1+C2] (folded BN scales multiplied into the rows) and bias = b1 + b2
 * from the two (Cout,C,1,1) weights and their folded scale/bias vectors.  C1, C2 % 32 == 0, Cout >= 128. */
int vatl_pack_conv1x1_dual_weight(const float* w1, const float* scale1, const float* bias1, const float* w2, const float* scale2,
                                  const float* bias2, float* out, float* bias, int Cout, int C1, int C2, int CoutPad, void* stream);
int vatl_conv1x1_dual_fwd(const float* a, const float* x, const float* w, const float* bias, float* y, int N, int Ho, int Wo,
                          int C1, int H2, int W2, int C2, int stride2, int Cout, int CoutPad, int relu, void* stream);

/* Winograd F(2x2,3x3) for 32 -> 32 channel 3x3 / stride 1 / pad 1 layers without cross-wave exchange (csrc/winograd_c32.hip; BasicBlock convs of HRNet's
 * highest-resolution branch, hrnet.py:24-56): x, residual, y NHWC (N,H,W,32), H and W even; u from vatl_pack_winograd_c32_weight ((32,32,3,3) filter ->
 * vatl_winograd_c32_weight_floats() floats); y = act(scale * conv(x) + bias + residual), scale / bias / residual may be NULL.  Same arithmetic as
 * vatl_conv3x3_winograd_fwd up to the order of the channel sums (fp32 rounding). */
int64_t vatl_winograd_c32_weight_floats(void);
int vatl_pack_winograd_c32_weight(const float* w, float* u, void* stream);
int vatl_conv3x3_winograd_c32_supported(int N, int H, int W, int Cin, int Cout);
int vatl_conv3x3_winograd_c32_fwd(const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y, int N, int H, int W,
                                  int relu, void* stream);

/* Winograd F(4x4,3x3) for 3x3 / stride 1 / pad 1 layers on grids of whole 4x4 tiles (csrc/winograd_f4.hip; Bottleneck.conv2 of ResNet stages 2 / 3, Resnet.py:104-128;
 * BasicBlock convs of HRNet's 128-channel branch, hrnet.py:24-56): 36 multiplies per (tile, input channel, output channel) for 16 outputs, 0.5625x the matrix work of
 * vatl_conv3x3_winograd_fwd.  x (N,H,W,Cin), residual / y (N,H,W,Cout) NHWC; u from vatl_pack_winograd_f4_weight ((Cout,Cin,3,3) filter -> vatl_winograd_f4_weight_floats
 * floats, U = G g G^T formed in float64); y = act(scale * conv(x) + bias + residual), scale / bias / residual may be NULL.  Served (vatl_conv3x3_winograd_f4_supported): H and W
 * multiples of 4, Cin >= 64 a multiple of 16, Cout a multiple of 64.  Exact fp32 products and sums like the F(2x2) route; the rounding differs in the last bits (tests hold both
 * routes to the same float64 bound), and a crop's bits do not depend on its batch position. */
int64_t vatl_winograd_f4_weight_floats(int Cout, int Cin);
/* (Cout, Cin) describe the PACKED filter; data_gradient: w is the forward filter (O = Cin, I = Cout, 3, 3) and u the filter of dX = conv(dY, rot180(w)^T). */
int vatl_pack_winograd_f4_weight(const float* w, float* u, int Cout, int Cin, int data_gradient, void* stream);
int vatl_conv3x3_winograd_f4_supported(int N, int H, int W, int Cin, int Cout);
int vatl_conv3x3_winograd_f4_fwd(const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y, int N, int H, int W,
                                 int Cin, int Cout, int relu, void* stream);
/* The same launch under `model.train()` (ActiveLearning.py:658-673), F(4x4,3x3) counterparts of vatl_conv3x3_winograd_fwd_stats / _fwd_bnbwd: _stats: y = conv(x) and the
 * per-(16-tile row block, channel) double (sum, sum of squares) partials of y in the layout vatl_bn_train_finalize reduces; _bnbwd (u packed with data_gradient = 1):
 * y = (conv(x) + residual) * [consumer layer's ReLU mask], `stats` the (sum g, sum g * xhat) partials.  Capacity vatl_winograd_f4_stats_row_blocks(N, H, W) * Cout * 2 doubles. */
int64_t vatl_winograd_f4_stats_row_blocks(int64_t N, int H, int W);
int vatl_conv3x3_winograd_f4_fwd_stats(const float* x, const float* u, float* y, double* stats, int64_t* row_blocks_used, int N, int H, int W, int Cin, int Cout,
                                       void* stream);
int vatl_conv3x3_winograd_f4_fwd_bnbwd(const float* x, const float* u, const float* residual, float* y, int N, int H, int W, int Cin, int Cout, const float* bn_z,
                                       const float* bn_mask_y, const float* bn_scale, const float* bn_bias, const float* bn_mean, const float* bn_invstd,
                                       double* stats, int64_t* row_blocks_used, void* stream);

/* 1x1 convolution with K = 128 (or, since round 6, 256) input channels as a row-streaming GEMM (csrc/conv1x1_rows.hip): y = act(scale * (A W^T) + bias + residual) for the short-K /
 * wide-N layers the tiled implicit GEMM runs far from both roofs (Bottleneck.conv3 of ResNet stage 2, Resnet.py:120-128; with x2: conv3 + projection
 * shortcut of stage 1's first block as vatl_conv1x1_dual_fwd computes it, Resnet.py:104-128, 185-189).  a (M, K1), x2 (M, K2) or NULL: the K columns
 * K1 .. K1 + K2 - 1 come from x2; w [N][K1 + K2] (vatl_pack_conv_weight layout of a 1x1 filter / vatl_pack_conv1x1_dual_weight); scale / bias / residual may
 * be NULL.  Bit-identical to vatl_conv2d_fwd / vatl_conv1x1_dual_fwd.  Served (vatl_conv1x1_rows_supported): K1 = 128, K2 = 0; K1 = K2 = 64;
 * K1 = 256, K2 = 0 (Bottleneck.conv3 of ResNet stage 3: 256 -> 1024); N a multiple of 128, <= 4096; (M + 32) * max(N, K) < 2^30. */
int vatl_conv1x1_rows_supported(int K1, int K2, int N, int64_t M);
int vatl_conv1x1_rows_fwd(const float* a, const float* x2, const float* w, const float* scale, const float* bias, const float* residual, float* y,
                          int64_t M, int K1, int K2, int N, int relu, void* stream);

/* Last 1x1 conv of a bottleneck (+ bn3 + identity skip + ReLU, Resnet.py:120-128) chained with the NEXT bottleneck's first 1x1 conv
 * (+ bn1 + ReLU, Resnet.py:104-108) in one launch (csrc/bottleneck_chain.hip): t = relu(scale3 * (a W3^T) + bias3 + skip) is written
 * (it is the next block's skip connection) and y1 = relu(scale1 * (t W1^T) + bias1) is computed from the tile still in LDS, so the
 * Cout-channel tensor is not re-read.  a (M, Cmid), skip / t (M, Cout), y1 (M, Cnext): NHWC pixels as rows; w3 [Cout][Cmid] and
 * w1 [Cnext][Cout] in the vatl_pack_conv_weight layout of 1x1 filters; scale / bias may be NULL (1 / 0), skip may be NULL.
 * Cnext == 0 (w1, y1 NULL): the first GEMM alone.  t has the bits of vatl_conv2d_fwd; y1 sums K in four fixed pieces.
 * vatl_bottleneck_chain_supported: 1 for the shapes served (Cmid 64, Cout 256, Cnext 64 or 0, (M + 32) * 256 < 2^30). */
int vatl_bottleneck_chain_supported(int Cmid, int Cout, int Cnext, int64_t M);
int vatl_bottleneck_chain_fwd(const float* a, const float* w3, const float* scale3, const float* bias3, const float* skip, float* t,
                              const float* w1, const float* scale1, const float* bias1, float* y1, int64_t M, int Cmid, int Cout,
                              int Cnext, void* stream);

/* Route selectors (benchmarks / A-B tests only).  PROCESS-GLOBAL (relaxed atomics inside the library; set them before
 * several host threads launch) and FROZEN: the shipped library accepts exactly the knobs of this table, and every accepted value
 * computes the SAME BITS as the default — a caller can never change results through this entry point (tests/test_gpu_conv.py and
 * tests/test_gpu_winograd.py assert the bit-identity knob by knob).  Anything else returns VATL_EINVAL.
 *
 *   knob  default  values          selects
 *    0      4      0, 2, 4, 5      k-loop schedule of the implicit-GEMM kernel (two-phase / interleaved / distance-2 prefetch / LDS-DMA)
 *    1      0      0, 1            tile order of the implicit-GEMM grid (n-tile or m-tile fastest)
 *    5      0      0, 64, 128      rows of the implicit-GEMM block tile (0 = chosen from the grid size).  Conv outputs are bit-identical;
 *                                  the float64 row-block partials of the training entry points' BatchNorm statistics follow the tile
 *    7      1      0 .. 64         persistent 1x1 GEMM kernel for K <= 256 v (0 = off)
 *    8      1      0, 1            halo-tile kernel for the 32-channel 3x3 layers (csrc/conv3x3_halo.hip) vs the implicit GEMM
 *   10      2      1, 2            operand look-ahead of the persistent 1x1 kernel in k-tiles
 *   18    2048     0 .. 2^20       KB of filter slices per group of the Winograd kernel's tile order (0 = one slice)
 *   21      2      1, 2, 3         32-channel filter halves per Winograd block (2 = two where the launch keeps >= 400 blocks, 3 = wherever possible)
 *   22      8      0 .. 4096       persistent Winograd route for layers of at most v 16-channel stages (0 = never)
 *   24      2      0 .. 3          prefetching variant of the persistent Winograd kernel (bit 0 / 1: one- / two-half blocks)
 *   25      1      0, 1            staging-address tables of the Winograd weight-gradient kernels (0 = addresses formed in the kernel)
 *
 * NOT in the shipped library: knobs that change the summation order (3 / 19 = pixel splits of the weight-gradient launches, 9 = split-K
 * cut policy, 12 = stream-K switch, 23 = gradient halves per Winograd weight-gradient block), performance-only experiments (2 = block
 * stagger, 16 = pixels per thread of the crop kernel) and the profiling ablations that compute WRONG results by construction (knob 0
 * values 10..13, knobs 4 / 6 / 17).  They exist only in the variant built with -DVATL_ABLATION (build.py --ablation ->
 * libvatl_hip_ablation.so, loaded through VATL_HIP_LIB by the tools under tools/); the wrong-result ones additionally need
 * VATL_ALLOW_ABLATION=1 in the environment of that variant. */
int vatl_tune_set(int knob, int value);
/* Opt-in split-K for small batches (single-frame / online inference): with a caller-owned workspace registered, conv
 * launches of fewer than 256 blocks and >= 16 k-tiles are cut along K into ~512 blocks; every split writes a raw partial
 * tile into its workspace slice and one pass sums the slices in order and applies scale / bias / residual / ReLU.
 * Deterministic, but the summation order differs from the unsplit kernel: results agree to fp32 rounding, not bit for bit,
 * and the choice depends on the batch size — so it is OFF unless a workspace is set (the evaluation path relies on
 * batch-size-independent bits).  The workspace is registered for the CURRENT device (launches on other devices never touch
 * it) and is shared by that device's streams: use from one stream at a time.  NULL disables it for the current device. */
int vatl_set_splitk_workspace(float* workspace, int64_t floats);
/* The same for the launches of the CALLING HOST THREAD only (thread-local; takes precedence over the device-wide workspace
 * while set; NULL clears it), always with the batch-invariant cut (the cut depends on the
 * layer's per-image geometry, so a crop's bits do not depend on how many crops share its call).  This is what module calls
 * with <= 16 crops use for their duration (scripts/poseestimator_eval.py shape): no global switch, no buffer shared between
 * host threads.  The workspace must be on the device the thread launches on; one stream per thread at a time. */
int vatl_set_splitk_workspace_thread(float* workspace, int64_t floats);
/* Stream-K for the fine-tune step's conv launches (forward and data gradient) of the CALLING HOST THREAD: while a workspace is
 * registered, a 64x128-tile launch that would keep fewer than 85 % of the chip's block slots busy is run as exactly 768
 * persistent blocks that share the flat (tile, k-tile) sequence evenly; a tile split between blocks is completed by the block
 * that holds its last k-tile, which adds the other blocks' raw accumulators from the workspace in a fixed order (bitwise
 * reproducible; agrees with the unsplit kernel to fp32 rounding, not bit for bit — which is why only the training paths use
 * it).  workspace: vatl_streamk_workspace_bytes() bytes of device memory, ZEROED once by the caller before it is registered
 * (flags carry a per-launch epoch and are never reset), owned by the caller, used by this thread's launches on one stream at
 * a time; NULL clears the registration. */
int64_t vatl_streamk_workspace_bytes(void);
int vatl_set_streamk_workspace_thread(void* workspace, int64_t bytes);
/* CoutPad the packer must use for a given Cout (multiple of the kernel's N tile). */
int vatl_conv_cout_pad(int Cout);
int vatl_conv2d_fwd(const float* x, const float* w, const float* scale, const float* bias, const float* residual,
                    float* y, int N, int H, int W, int Cin, int Cout, int CoutPad, int R, int S, int stride, int pad,
                    int relu, int out_nchw, void* stream);

/* ConvTranspose2d(4,2,1,bias=False)+BatchNorm2d+ReLU as four 2x2 sub-pixel
 * convolutions (no zero stuffing).  x (N,H,W,Cin) -> y (N,2H,2W,Cout) NHWC.
 * simplepose.py:37-59, :84. */
int vatl_deconv4x4s2_fwd(const float* x, const float* w, const float* scale, const float* bias, float* y,
                         int N, int H, int W, int Cin, int Cout, int CoutPad, int relu, void* stream);

/* MaxPool2d(3,2,1) on NHWC (Resnet.py:158,172).  C % 4 == 0. */
int vatl_maxpool3x3s2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* AdaptiveAvgPool2d(1)+flatten on NHWC -> (N,C)  (simplepose.py:88-91). */
int vatl_gap_fwd(const float* x, float* y, int N, int HW, int C, void* stream);

/* nn.PixelShuffle(2) on NHWC: (N,H,W,C) -> (N,2H,2W,C/4), out[2y+i][2x+j][c] = in[y][x][4c+2i+j]
 * (fastpose.py:42,56; DUC.py:23,28).  C % 16 == 0. */
int vatl_pixelshuffle2_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* SE gate + residual + ReLU, NHWC: y = relu(x * sigmoid(gate[n][c]) + residual)
 * (SE_module.py:20-24 with the second Linear's pre-sigmoid output as `gate`; SE_Resnet.py:125-135). */
int vatl_se_scale_add_relu(const float* x, const float* gate, const float* residual, float* y, int N, int HW, int C, void* stream);
/* HRNet branch fusion, NHWC: y = act(base + sum_k nearest_upsample(z_k, 2^shift_k)); z_k is
 * (N, H>>shift_k, W>>shift_k, C) or NULL (hrnet.py:199-203 Upsample(nearest), :242-260 sum + ReLU). */
int vatl_fuse_upsample_add(const float* base, const float* z0, int shift0, const float* z1, int shift1, const float* z2, int shift2,
                           float* y, int N, int H, int W, int C, int relu, void* stream);
/* Backward of the nearest up-sampling term of that fusion: dz (N,H>>shift,W>>shift,C) = block sums of
 * dy * [yact > 0] over 2^shift x 2^shift pixels (yact = the fused output, NULL = unmasked); autograd of hrnet.py:199-203. */
int vatl_upsample_nearest_bwd(const float* dy, const float* yact_or_null, float* dz, int N, int H, int W, int C, int shift, void* stream);
/* Backward of AdaptiveAvgPool2d(1) (simplepose.py:88-91, SE_module.py:21): dx[n][p][c] = dy[n][c] / HW. */
int vatl_gap_bwd(const float* dy, float* dx, int N, int HW, int C, void* stream);

/* ------------------------------------------------------------------------ *
 * Scorers on heat-maps (N,J,H,W) fp32 NCHW, one pass over HBM each
 * ------------------------------------------------------------------------ */

/* heatmap_to_coord_simple (transforms.py:550-583): first-max arg-max, +-0.25 px
 * shift, inverse crop affine (float64, float32-rounded control points).
 * bbox (N,4) xyxy fp32; coords (N,J,2) fp32 image px; maxvals (N,J) fp32;
 * idx (N,J) int32 flat arg-max (may be NULL). */
int vatl_decode_argmax_affine(const float* hm, const float* bbox, float* coords, float* maxvals, int32_t* idx,
                              int N, int J, int H, int W, void* stream);
/* The same decode writing the reference's key-point rows directly — kpts (N,J,3) fp32 = (x, y, score) per joint, the
 * np.concatenate((pose_coords, pose_scores), axis=1) of ActiveLearning.py:304-306 — plus the two per-item scores made from
 * them: hp (N) = -np.sum(pose_scores) (the 'HP' uncertainty, :329-330; NumPy's float32 pairwise order, bit-identical) and
 * pose_score (N doubles) = float(np.mean(s) + 1.25 * np.max(s)) (the json "score", :314) with the promotion of the reference's pinned
 * numpy 1.23.5: float32 mean, float64 product and sum (python float x NumPy scalar -> float64).  idx, hp, pose_score may be NULL. */
int vatl_decode_pose(const float* hm, const float* bbox, float* kpts, int32_t* idx, float* hp, double* pose_score,
                     int N, int J, int H, int W, void* stream);

/* compute_thc (ActiveLearning.py:747-760) for P pairs: out[i] = sum|a_i-b_i|/J
 * (norm 1) or sum (a_i-b_i)^2/J (norm 2); a_i = a + i*stride_a, b likewise
 * (strides in floats; stride J*H*W and b = a + J*H*W gives consecutive frames). */
int vatl_thc_pairs(const float* a, const float* b, int64_t stride_a, int64_t stride_b, float* out,
                   int P, int J, int HW, int norm, void* stream);
/* The isPrev/isNext rule of eval_and_query (ActiveLearning.py:345-363) on a
 * de-duplicated id-sorted stream: thc[i] = [prev]*pair[i-1] + [next]*pair[i],
 * doubled when exactly one neighbour exists.  pair has N-1 entries. */
int vatl_thc_combine(const float* pair, const uint8_t* is_prev, const uint8_t* is_next, float* thc, int N, void* stream);

/* localpeak_mean (local_peak.py:5-22): per item the mean of 3x3 zero-padded
 * local maxima >= order * largest local maximum, pooled over joints (nan when
 * none).  count (N,J) int32 kept peaks per joint (may be NULL).  workspace: 2*N*J doubles (per-plane sum and count;
 * one block per plane, the per-item mean is formed in joint order by a second launch). */
int vatl_localpeak_mean(const float* hm, float* mean, int32_t* count, double* workspace, int N, int J, int H, int W, float order,
                        void* stream);

/* compute_hybrid + WholeBodyAE + MSELoss (hybrid_feature.py:14-59,
 * AutoEncoder.py:13-39, ActiveLearning.py:364-386).  kpts (N,17,3) (x,y,score)
 * fp32, bbox (N,4) crop box xyxy; ae = 8 (weight,bias) pairs concatenated in
 * state-dict order (encoder.0,2,4,6, decoder.0,2,4,6), D in {38,42}, z = code
 * width; only38 != 0 drops feature indices 3,4,20,21 before the MSE (D must be
 * 42).  wpu (N) fp32; status (N) int32: 0 ok, 1 bbox height <= 0, 2 score sum
 * <= 0 (the reference's two asserts). */
int vatl_hybrid_ae_wpu(const float* kpts, const float* bbox, const float* ae, int D, int z, int only38,
                       float* wpu, int32_t* status, int N, void* stream);

/* TPC (ActiveLearning.py:333-344, compute_tpc :736-745) on a de-duplicated id-sorted
 * stream: cur (N,J,2) decoded key-points; adj_prev[i] / adj_next[i] (N,J,2) = heat-maps
 * of items i-1 / i+1 decoded with item i's box (vatl_decode_argmax_affine on shifted
 * views); counts joints displaced by more than 0.01*sqrt(box area), summed over the
 * existing neighbours, doubled when exactly one exists.  tpc (N) fp32 (integer valued). */
int vatl_tpc_stream(const float* cur, const float* adj_prev, const float* adj_next, const float* bbox,
                    const uint8_t* is_prev, const uint8_t* is_next, float* tpc, int N, int J, void* stream);

/* heatmap_to_coord_simple_regress (transforms.py:586-702; LOSS.TYPE L1JointRegression):
 * soft-arg-max expectation + the same inverse crop affine.  norm_type 0 softmax,
 * 1 sigmoid, 2 divide_sum.  coords (N,J,2), scores (N,J) (1, max sigmoid, 1). */
int vatl_decode_softargmax(const float* hm, const float* bbox, float* coords, float* scores,
                           int N, int J, int H, int W, int norm_type, void* stream);

/* WholeBodyAE.forward on given features (AutoEncoder.py:36-39): feat (N,D) -> recon
 * (N,D) (may be NULL) and per-item mean squared reconstruction error mse (N) (may be
 * NULL); ae packed as for vatl_hybrid_ae_wpu; D, z in 1..64. */
int vatl_ae_forward(const float* feat, const float* ae, int D, int z, float* recon, float* mse, int N, void* stream);

/* compute_hybrid (hybrid_feature.py:14-59) on float64 inputs: kpts (N,51), bbox (N,4)
 * as (x,y,w,h) -> feat (N,42) float64; status (N) as in vatl_hybrid_ae_wpu (may be NULL). */
int vatl_hybrid_feature_f64(const double* kpts, const double* bbox_xywh, double* feat, int32_t* status, int N, void* stream);

/* localpeak_values (local_peak.py:5-10) as a mask: mask[p][y][x] = 1 where the pixel is a
 * kept local peak of plane p (planes = N*J maps of H x W). */
int vatl_localpeak_mask(const float* hm, uint8_t* mask, int planes, int H, int W, float order, void* stream);

/* Multiple-peak criteria (ActiveLearning.py:762-788): per (item, joint) plane the <= 5 local peaks that
 * skimage.feature.peak_local_max(plane, min_distance, num_peaks=5) returns (values, flat indices y*W+x or -1, count),
 * the plane's MPE term entropy(softmax(peak values)) and its Margin term |peak0 - peak1| (0 with fewer than two
 * peaks).  compute_mpe / compute_margin are the sums of those terms over the joints of an item.
 * peak_val, peak_idx: N*J*5; npeaks, mpe, margin: N*J. */
int vatl_peaks5(const float* hm, float* peak_val, int32_t* peak_idx, int32_t* npeaks, float* mpe, float* margin,
                int N, int J, int H, int W, int min_distance, void* stream);
/* compute_entropy (ActiveLearning.py:790-796): out[n*J + j] = scipy.stats.entropy(plane.flatten()) — normalised by the
 * plane sum, -inf for a negative entry, nan for a zero sum, exactly as scipy returns them. */
int vatl_plane_entropy(const float* hm, float* out, int N, int J, int H, int W, void* stream);

/* ------------------------------------------------------------------------ *
 * Training-mode backbone (ActiveLearning.py:658-673: model.train(), loss.backward())
 * ------------------------------------------------------------------------ */

/* Generalised forward conv behind the data-gradient paths: explicit GEMM pixel grid Ho x Wo,
 * separate paddings, and an output scatter (oy*osy+ooy, ox*osx+oox) into an OH x OW NHWC image
 * (dgrad of a strided conv = per-parity launches over the output-gradient grid). Cin % 32 == 0. */
int vatl_conv2d_fwd_ex(const float* x, const float* w, const float* scale, const float* bias, const float* residual, float* y,
                       int N, int H, int W, int Cin, int Cout, int CoutPad, int R, int S, int stride, int pad_y, int pad_x,
                       int Ho, int Wo, int OH, int OW, int osy, int osx, int ooy, int oox, int relu, void* stream);
/* The same launch fused with the reduction pass of the BatchNorm backward of the layer whose output gradient it produces
 * (loss.backward() through Conv2d -> BatchNorm2d -> ReLU chains, ActiveLearning.py:672; Resnet.py:104-128): y receives
 * g = (W^T dz + residual) * mask, mask = [bn_mask_y > 0] when bn_mask_y is given (ReLU after a residual sum), else
 * [fmaf(bn_z, bn_scale, bn_bias) > 0] when bn_scale is given (plain Conv+BN+ReLU, the mask vatl_scale_bias_act produced), else 1;
 * `stats` receives per (row block of the implicit GEMM, channel) the double partials (sum g, sum g*xhat), xhat =
 * (bn_z - bn_mean)*bn_invstd, in the layout of vatl_conv2d_fwd_stats (capacity vatl_conv_stats_row_blocks(N*Ho*Wo, 1) * Cout * 2
 * doubles per launch; *row_blocks_used = HOST count actually written).  bn_z / bn_mask_y have the layout of y.  Cout % 4 == 0.
 * vatl_bn_bwd_from_stats then finishes the BatchNorm backward with one pass: no separate reduction over (dy, z). */
int vatl_conv2d_fwd_ex_bnbwd(const float* x, const float* w, const float* residual, float* y, int N, int H, int W, int Cin, int Cout,
                             int CoutPad, int R, int S, int stride, int pad_y, int pad_x, int Ho, int Wo, int OH, int OW, int osy, int osx,
                             int ooy, int oox, const float* bn_z, const float* bn_mask_y, const float* bn_scale, const float* bn_bias,
                             const float* bn_mean, const float* bn_invstd, double* stats, int64_t* row_blocks_used, void* stream);
/* Data-gradient weights of nn.Conv2d (Cout,Cin,R,S): out[c][t][n] = w[n][c][tap_r[t]][tap_s[t]],
 * shape [CinPad][ntaps][CoutK] (rows c >= Cin and columns n >= Cout zero); tap_r/tap_s are HOST arrays,
 * ntaps <= 16. */
int vatl_pack_dgrad_weight(const float* w_oihw, float* out, int Cout, int Cin, int R, int S, int CinPad, int CoutK,
                           int ntaps, const int* tap_r, const int* tap_s, void* stream);
/* Weight gradient of nn.Conv2d: dw (Cout,Cin,R,S) = sum over pixels of dz (x) x.  x NHWC (N,H,W,Cin)
 * (Cin = 3 means the 4-channel padded stem input), dz NHWC (N,Ho,Wo,CoutG) with channel stride
 * CoutG >= Cout.  fp32 MFMA; the pixel range is split over blocks, every split writes a partial gradient into its slice
 * of the workspace (vatl_conv2d_wgrad_workspace_floats(Cout, Cin, R, S, M = N*Ho*Wo) floats) and one pass sums the slices
 * in split order while restoring the OIHW layout: no atomics, bitwise reproducible. */
int64_t vatl_conv2d_wgrad_workspace_floats(int Cout, int Cin, int R, int S, int64_t M);
int vatl_conv2d_wgrad(const float* x, const float* dz, float* dw, float* workspace, int N, int H, int W, int Cin,
                      int Cout, int CoutG, int R, int S, int stride, int pad, void* stream);
/* Weight gradient of nn.ConvTranspose2d(4,2,1): dw (Cin,Cout,4,4); x NHWC (N,H,W,Cin), dy NHWC (N,2H,2W,Cout). */
int64_t vatl_deconv4x4s2_wgrad_workspace_floats(int Cin, int Cout, int64_t M /* = N*H*W input pixels */);
int vatl_deconv4x4s2_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int H, int W, int Cin,
                           int Cout, void* stream);

/* BatchNorm2d in training mode on NHWC rows (M = N*H*W, C): batch mean / biased variance ->
 * save_mean, save_invstd and the affine (scale = gamma*invstd, bias = beta - mean*scale); running stats
 * updated in place with `momentum` (unbiased variance), NULL to skip.  workspace:
 * vatl_col_reduce_workspace_doubles(M, C) doubles.  (Resnet.py:67 etc. under model.train()) */
int64_t vatl_col_reduce_workspace_doubles(int64_t M, int C);
int vatl_bn_train_fwd_stats(const float* z, int64_t M, int C, const float* gamma, const float* beta, float* running_mean,
                            float* running_var, float momentum, float eps, float* save_mean, float* save_invstd,
                            float* scale, float* bias, double* workspace, void* stream);
/* y = act(z*scale[c] + bias[c] (+ residual)) on NHWC rows; C % 4 == 0. */
int vatl_scale_bias_act(const float* z, const float* scale, const float* bias, const float* residual, float* y,
                        int64_t M, int C, int relu, void* stream);
/* Backward of BN(train)+ReLU: g = dy*[y>0] (y NULL: no ReLU), dbeta = sum g, dgamma = sum g*xhat,
 * dz = gamma*invstd*(g - dbeta/M - xhat*dgamma/M); g_out (may be NULL) receives g (the gradient of a
 * residual input).  coef3C: 3*C floats scratch; workspace as above. */
int vatl_bn_train_bwd(const float* dy, const float* y_or_null, const float* z, const float* gamma, const float* save_mean,
                      const float* save_invstd, float* dz, float* g_out_or_null, float* dgamma, float* dbeta,
                      int64_t M, int C, float* coef3C, double* workspace, void* stream);
/* The same backward for a Conv+BN+ReLU layer WITHOUT a skip input: the ReLU mask is recomputed from z as
 * fmaf(z, scale, bias) > 0 (bit-identical to what vatl_scale_bias_act stored), so y is not read. */
int vatl_bn_train_bwd_relu(const float* dy, const float* scale, const float* bias, const float* z, const float* gamma,
                           const float* save_mean, const float* save_invstd, float* dz, float* dgamma, float* dbeta,
                           int64_t M, int C, float* coef3C, double* workspace, void* stream);
/* BatchNorm backward from the partial sums vatl_conv2d_fwd_ex_bnbwd left behind: g = the masked output gradient it stored,
 * partial / row_blocks = its statistics (several launches may have appended theirs: per-parity data gradients of a strided
 * conv).  dgamma = sum g*xhat, dbeta = sum g, dz = gamma*invstd*(g - dbeta/M - xhat*dgamma/M) in one pass over (g, z).
 * coef3C: 3*C floats of scratch. */
int vatl_bn_bwd_from_stats(const double* partial, int64_t row_blocks, const float* g, const float* z, const float* gamma,
                           const float* save_mean, const float* save_invstd, float* dz, float* dgamma, float* dbeta, int64_t M, int C,
                           float* coef3C, void* stream);
/* Training forward with the BatchNorm batch statistics taken in the conv epilogue (no separate pass over z):
 * the conv / deconv writes z (no affine, no ReLU) and, per 128-row block of the implicit GEMM and channel, a partial
 * (sum, sum of squares) pair into `stats` (capacity vatl_conv_stats_row_blocks(rows, phases) * Cout * 2 doubles; rows =
 * N*Ho*Wo, phases = 1, or N*H*W and 4 for the transposed conv; the number of row blocks actually written — it depends
 * on the tile the launch picked — is returned in the HOST variable *row_blocks_used and is what vatl_bn_train_finalize
 * takes as `row_blocks`).  vatl_bn_train_finalize reduces the partials in a fixed order
 * and produces what vatl_bn_train_fwd_stats produces (nn.BatchNorm2d under model.train(), Resnet.py:104-128 via
 * ActiveLearning.py:658-672); M = elements per channel (N*Ho*Wo of the layer OUTPUT). */
int64_t vatl_conv_stats_row_blocks(int64_t gemm_rows, int phases);
int vatl_conv2d_fwd_stats(const float* x, const float* w, float* y, double* stats, int64_t* row_blocks_used, int N, int H, int W,
                          int Cin, int Cout, int CoutPad, int R, int S, int stride, int pad, void* stream);
int vatl_deconv4x4s2_fwd_stats(const float* x, const float* w, float* y, double* stats, int64_t* row_blocks_used, int N, int H, int W,
                               int Cin, int Cout, int CoutPad, void* stream);
int vatl_bn_train_finalize(const double* partial, int64_t row_blocks, int64_t M, int C, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* save_mean,
                           float* save_invstd, float* scale, float* bias, void* stream);
/* MaxPool2d(3,2,1) backward on NHWC: x (N,H,W,C) forward input, dy (N,Ho,Wo,C) -> dx. */
int vatl_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* Training-mode max-pool: forward that also records the winning tap (0..8, first maximum) per pooled element,
 * and the backward that consumes it.  idx (N,Ho,Wo,C) uint8. */
int vatl_maxpool3x3s2_fwd_idx(const float* x, float* y, uint8_t* idx, int N, int H, int W, int C, void* stream);
int vatl_maxpool3x3s2_bwd_idx(const float* dy, const uint8_t* idx, float* dx, int N, int H, int W, int C, void* stream);
/* The stem tail bn1 -> relu -> maxpool of the ResNet trunks under model.train() (Resnet.py:171-172 via ActiveLearning.py:658-672)
 * without the full-resolution activation or its gradient in HBM.  Forward: y (N,Ho,Wo,C) = maxpool(relu(z*scale + bias)) with
 * the affine + ReLU applied to the conv output z on load (the fmaf of vatl_scale_bias_act: identical values and winners), idx
 * as above.  Backward: dpool = gradient of y; the gradient of every stem pixel is gathered from the <= 4 windows covering it
 * inside the BatchNorm backward's reduction and apply passes: dz (N,H,W,C), dgamma, dbeta as vatl_bn_train_bwd_relu.
 * C % 4 == 0; workspace: vatl_col_reduce_workspace_doubles(N*H*W, C) doubles; coef3C: 3*C floats. */
int vatl_maxpool3x3s2_fwd_idx_affine(const float* z, const float* scale, const float* bias, float* y, uint8_t* idx, int N, int H, int W, int C,
                                     void* stream);
int vatl_bn_train_bwd_relu_pool(const float* dpool, const uint8_t* idx, const float* scale, const float* bias, const float* z,
                                const float* gamma, const float* save_mean, const float* save_invstd, float* dz, float* dgamma,
                                float* dbeta, int N, int H, int W, int C, float* coef3C, double* workspace, void* stream);
/* Every weight re-pack of one fine-tune step in ONE launch (the per-tensor entry points vatl_pack_conv_weight,
 * vatl_pack_dgrad_weight, vatl_pack_deconv4x4s2_weight produce the same bytes one launch each).  jobs_device: device
 * array of njobs descriptors sorted by first_block = the running sum of the jobs' block counts over the preceding jobs — kinds 0 / 2:
 * ceil(elements / 1024); kind 1: ceil(a / 32) * c * ceil(b / 32) (one 32 x 32 tile of one tap per block); kinds 3 .. 6: elements / 4096;
 * total_blocks = that sum over all jobs.  kind 0: conv forward layout, (a, b, c) = (CoutPad, Spad, CinPad);
 * kind 1: data-gradient layout, (a, b, c) = (CinPad, CoutK, ntaps) with the taps in tap_r / tap_s;
 * kind 2: ConvTranspose2d(4,2,1) layout, src (Cin,Cout,4,4), a = CoutPad;
 * kind 5: Winograd F(3x3,2x2) phase filters of ConvTranspose2d(4,2,1) (vatl_pack_winograd_deconv_weight): src (Cin,Cout,4,4), a / b as
 *   for kinds 3 / 4, c = Cout;  kind 6: the filters of its data gradient (vatl_pack_winograd_deconv_dgrad_weight): (Cout, Cin) fields =
 *   (layer Cin, layer Cout), c = layer Cout;
 * kind 7 / 8: Winograd F(4x4,3x3) filter transform, forward / data gradient (vatl_pack_winograd_f4_weight): (Cout, Cin) of the PACKED filter, c = inner dimension
 *   of src; ceil(elements / 1024) blocks.
 * kind 3 / 4: Winograd F(2x2,3x3) filter transform, forward / data gradient (vatl_pack_winograd_weight): (Cout, Cin) of the PACKED
 *   filter, a = its padded Cout, b = 32-channel groups per tile (a / 32 <= 1 ? 1 : 2), c = Cin (kind 3) or Cout (kind 4) = the inner
 *   dimension of src. */
typedef struct VatlPackJob {
    const float* src;
    float* dst;
    int64_t first_block;
    int32_t kind;
    int32_t Cout, Cin, R, S;
    int32_t a, b, c;
    int32_t tap_r[16], tap_s[16];
} VatlPackJob;
int vatl_pack_weights_multi(const VatlPackJob* jobs_device, int njobs, int64_t total_blocks, void* stream);
/* Backward of nn.PixelShuffle(2) on NHWC: x (N,2H,2W,C/4) -> y (N,H,W,C). */
int vatl_pixelunshuffle2(const float* x, float* y, int N, int H, int W, int C, void* stream);
/* Backward of the SE-gated residual y = relu(u*sigmoid(gate) + shortcut) (SE_Resnet.py:125-135), two stages:
 *  1 (dgate != NULL): dgate[n][c] = sigmoid'(gate) * sum_hw dy*[y>0]*u  — gradient of the second Linear's output;
 *  2 (du != NULL):    gm = dy*[y>0] (shortcut gradient), du = gm*sigmoid(gate) + dpool[n][c]/HW where dpool is the
 *                     gradient that came back through the two Linear layers to the average-pooled vector. */
int vatl_se_bwd(const float* dy, const float* y, const float* u, const float* gate, const float* dpool_or_null,
                float* dgate_or_null, float* du_or_null, float* gm_or_null, int N, int HW, int C, void* stream);
/* dx = dy * [y > 0] on a flat fp32 span (ReLU backward of the SE MLP). */
int vatl_relu_bwd(const float* dy, const float* y, float* dx, int64_t n, void* stream);
/* out[c] = sum over rows of x (M,C)  (conv bias gradient). */
int vatl_col_sum(const float* x, int64_t M, int C, float* out, double* workspace, void* stream);

/* ------------------------------------------------------------------------ *
 * Fine-tune step pieces (ActiveLearning.py:662-677)
 * ------------------------------------------------------------------------ */

/* loss = 0.5*mean((o*m - t*m)^2) and dL/do; mask (N,J).  partial: workspace of
 * vatl_masked_mse_workspace_floats(N*J*HW) floats; loss: 1 float on device. */
int64_t vatl_masked_mse_workspace_floats(int64_t numel);
int vatl_masked_mse_fwd_bwd(const float* out, const float* target, const float* mask, float* grad, float* loss,
                            float* partial, int N, int J, int HW, void* stream);

/* L1JointRegression (alphapose/models/criterion.py:46-94: soft-arg-max "integral" regression with the symmetric
 * +-2 gradient of IngetralCoordinate :13-43, weighted L1, / batch when size_average): loss (1 float), grad of the
 * loss w.r.t. the heat-maps (B,J,H,W), the predicted normalised joints pred_jts (B, 2J) in [-0.5, 0.5).
 * gt_joints / gt_joints_vis: (B, 2J).  norm_type 0 softmax, 1 sigmoid, 2 divide_sum.  partial: B*J doubles. */
int vatl_l1_joint_regression_fwd_bwd(const float* hm, const float* gt_joints, const float* gt_joints_vis, float* grad, float* loss,
                                     float* pred_jts, double* partial, int B, int J, int H, int W, int norm_type, int size_average,
                                     void* stream);

/* SimpleTransform._target_generator (alphapose/utils/presets/simple_transform.py:122-158) for a batch: joints_xy (N,J,2)
 * input-pixel coordinates, vis (N,J) -> target (N,J,H,W) with an un-normalised Gaussian (sigma, radius 3 sigma, clipped
 * at the borders) centred on int(x / stride + 0.5), weight (N,J) = vis, zeroed when the patch misses the map; the
 * patch is drawn only for weight > 0.5.  Like the reference (:130-131) x is divided by in_h / H and y by in_w / W — the
 * two strides are equal (4) for every preset. */
int vatl_gaussian_targets(const float* joints_xy, const float* vis, float* target, float* weight, int N, int J, int H, int W,
                          int in_h, int in_w, float sigma, void* stream);

/* Crop producer, the step before the backbone (SimpleTransform.test_transform / __call__,
 * alphapose/utils/presets/simple_transform.py:81-98, 179-251): cv2.warpAffine(img, trans, (out_w, out_h), INTER_LINEAR)
 * [OpenCV 4.8 fixed-point bilinear, constant border 0] -> im_to_torch (alphapose/utils/transforms.py:76-91: CHW fp32,
 * / 255 only when the crop's maximum exceeds 1) -> minus mean per channel (simple_transform.py:93-95).
 * arena: packed uint8 frames, each (h, w, 3) row-major; per crop b: src_off[b] = byte offset of its frame in arena,
 * src_hwf[b] = {h, w, mirror} (mirror = 1 reads the frame flipped left-right like img[:, ::-1, :], :224),
 * minv[b] = the 6 doubles of the dst -> src map cv::warpAffine gets by inverting `trans` (source coordinates must stay
 * below 2^20 px).  out: (B, 3, out_h, out_w) fp32.  crop_max: B int32 of workspace (holds each crop's u8 maximum on
 * return).  B <= 65535, out_w <= 4096, out_h * out_w < 2^20. */
int vatl_crop_warp_affine(const uint8_t* arena, const int64_t* src_off, const int32_t* src_hwf, const double* minv, float* out,
                          int32_t* crop_max, int B, int out_h, int out_w, float mean0, float mean1, float mean2, void* stream);

/* One fine-tune step of the WholeBodyAE (ActiveLearning.py:905-925: AE forward, MSELoss(output, input), backward,
 * torch.optim.Adam) on a mini-batch feat (B, D), B <= 12 (the reference uses 10), in one launch.  ae / m / v: the packed parameters
 * (state-dict order W0,b0,...,W7,b7 = what vatl_hybrid_ae_wpu reads) and the Adam moments, updated in place;
 * `step` is the 1-based Adam step; loss (1 float, before the update) may be NULL. */
int vatl_ae_train_step(float* ae, float* m, float* v, const float* feat, int B, int D, int z, double lr, double beta1, double beta2,
                       double eps, int step, float* loss_or_null, void* stream);

/* torch.optim.AdamW step on one flat fp32 span (decoupled weight decay);
 * hyper-parameters are doubles like the Python floats torch derives its
 * per-step scalars from; `step` is the 1-based step count. */
int vatl_adamw_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                    double eps, double weight_decay, int step, void* stream);

/* The same update for a whole parameter group in one launch: table_dev = n_tensors rows of {p, g, m, v, numel, first_block}
 * (device pointers, the element count, and the running sum of ceil(numel / vatl_adamw_multi_block_elems()) over the preceding
 * rows — all int64, resident on the device); total_blocks = that sum over all rows.  Every tensor shares the hyper-parameters
 * and the step count; tensors get blocks in proportion to their size.  Arithmetic per element identical to vatl_adamw_step. */
int vatl_adamw_step_multi(const int64_t* table_dev, int n_tensors, int64_t total_blocks, double lr, double beta1, double beta2, double eps,
                          double weight_decay, int step, void* stream);
int64_t vatl_adamw_multi_block_elems(void);

/* torch.optim.Adam step (ActiveLearning.py:222-223): like AdamW but `weight_decay`
 * is an L2 term added to the gradient (the reference passes none: 0). */
int vatl_adam_step(float* p, const float* g, float* m, float* v, int64_t n, double lr, double beta1, double beta2,
                   double eps, double weight_decay, int step, void* stream);

/* torch.optim.SGD step with momentum (dampening 0, no Nesterov), ActiveLearning.py:220-221
 * (momentum 0.9, weight_decay 5e-4): g' = g + wd*p; buf = g' on step 1, momentum*buf + g' after;
 * p -= lr*buf. */
int vatl_sgd_step(float* p, const float* g, float* buf, int64_t n, double lr, double momentum, double weight_decay, int step,
                  void* stream);

/* compute_OKS (al_metric.py:42-69; called per item at ActiveLearning.py:309): object key-point similarity of the decoded
 * key-points pred (N,17,3) fp32 against ground truth gt (N,51) float64 with the annotation box (N,4) xywh float64 as the
 * object scale; invisible-everywhere items fall back to the distance from the doubled box.  out (N) float64. */
int vatl_oks(const float* pred_kpts, const double* gt_kpts, const double* bbox_xywh, double* out, int N, void* stream);

/* ------------------------------------------------------------------------ *
 * Query selection on the (N, D) fp32 embeddings; float64 accumulation like the
 * reference's float64 fvecs_matrix.
 * ------------------------------------------------------------------------ */
/* out[i] = sum_j cosine_distance(x_i, x_j): the row sums of sklearn's KNeighborsTransformer(mode='distance',
 * metric='cosine', n_neighbors=n-1) matrix used as "influence" / "diversity" score (ActiveLearning.py:467-476,
 * 583-592).  workspace: n + D doubles. */
int vatl_cosine_rowsum(const float* emb, int64_t n, int D, double* out, double* workspace, void* stream);
/* k-center greedy (ActiveLearning.py:798-850).  update: min_dist[i] = min(min_dist[i], ||x_i - x_c||) over the
 * centres in the DEVICE index array (first != 0: min_dist is write-only);  pick: selected_dev[step] = first arg-max of
 * a*min_dist + b*unc (either may be NULL), then unc[selected] = 0 — so pick -> update(selected_dev + step, 1) chains on
 * the stream without host round trips. */
int vatl_kcenter_update(const float* emb, int64_t n, int D, const int32_t* centers_dev, int n_centers, double* min_dist, int first,
                        void* stream);
int vatl_kcenter_pick(const double* min_dist_or_null, double* unc_or_null, double a, double b, int32_t* selected_dev, int step,
                      int64_t n, void* stream);

/* ---- Winograd F(2x2, 3x3) route of the 3x3 / stride 1 / pad 1 layers (csrc/conv_winograd.hip) ------------------------------------
 * Replaces, for those layers, what the reference runs as torch.nn.Conv2d(k=3, s=1, p=1) inside Bottleneck / BasicBlock
 * (alphapose/models/layers/Resnet.py:52-78 conv2, alphapose/models/hrnet.py:24-56) — the reference's cuDNN picks the same algorithm
 * for them.  fp32 products and accumulation; 2.25x fewer multiplies than the direct sum, rounding differs in the last bits.
 *
 * vatl_pack_winograd_weight: w = the layer's (Cout, Cin, 3, 3) filter (OIHW, contiguous); u receives G g G^T in MFMA fragment order,
 *   vatl_winograd_weight_floats(Cout, Cin) floats (Cout padded to vatl_winograd_cout_pad).  data_gradient != 0: w is still the FORWARD
 *   filter (O, I, 3, 3) and (Cout, Cin) = (I, O): u is the filter of dX = conv(dY, rot180(w)^T).  Cin % 16 == 0.
 * vatl_conv3x3_winograd_fwd: y = act((x * w) * scale + bias + residual), x / y / residual NHWC fp32, Cout % 4 == 0.
 * vatl_conv3x3_winograd_fwd_stats: y = x * w and the per-(row block, channel) double (sum, sum of squares) partials of y in the layout
 *   vatl_bn_train_finalize reduces; capacity vatl_winograd_stats_row_blocks(N, H, W) * Cout * 2 doubles. */
int vatl_winograd_cout_pad(int Cout);
int64_t vatl_winograd_weight_floats(int Cout, int Cin);
int vatl_pack_winograd_weight(const float* w, float* u, int Cout, int Cin, int data_gradient, void* stream);
int vatl_conv3x3_winograd_fwd(const float* x, const float* u, const float* scale, const float* bias, const float* residual, float* y,
                              int N, int H, int W, int Cin, int Cout, int relu, void* stream);
/* Which kernel the calling thread's last vatl_conv3x3_winograd_fwd launch took: 0 = one block per (32-tile group, filter tile);
 * 1 = the persistent route (blocks keep a group of the period lcm(tiles per image, 32) and walk whole periods: the layers with
 * <= 128 input channels, whose blocks are otherwise shorter than their own set-up); 2 = persistent + a plain launch for the
 * images that do not fill a period.  All three give the same bits; tests use this to prove which one ran. */
int vatl_winograd_last_route(void);
int64_t vatl_winograd_stats_row_blocks(int64_t N, int H, int W);
int vatl_conv3x3_winograd_fwd_stats(const float* x, const float* u, float* y, double* stats, int64_t* row_blocks_used, int N, int H, int W,
                                    int Cin, int Cout, void* stream);
/* Winograd counterpart of vatl_conv2d_fwd_ex_bnbwd (3x3 / stride 1 / pad 1 data gradients; u packed with data_gradient = 1): y receives
 * g = (conv(x) + residual) * [consumer layer's ReLU mask], `stats` the (sum g, sum g * xhat) row-block partials
 * (capacity vatl_winograd_stats_row_blocks(N, H, W) * Cout * 2 doubles). */
/* ConvTranspose2d(4, 2, 1) on the Winograd route: each of the four sub-pixel phases is a 2x2 convolution of the input, run as
 * F(3x3, 2x2) (16 multiplies per 3x3 tile of phase outputs instead of 36).  Replaces nn.ConvTranspose2d(k=4, s=2, p=1) of
 * simplepose.py:46-66 (_make_deconv_layer) for Cin % 16 == 0, Cout % 4 == 0.  w = (Cin, Cout, 4, 4) as torch stores it; u receives
 * vatl_winograd_deconv_weight_floats(Cout, Cin) floats.  y: N x 2H x 2W x Cout (NHWC).  _stats: training forward, capacity
 * vatl_winograd_deconv_stats_row_blocks(N, H, W) * Cout * 2 doubles. */
int64_t vatl_winograd_deconv_weight_floats(int Cout, int Cin);
int vatl_pack_winograd_deconv_weight(const float* w, float* u, int Cout, int Cin, void* stream);
int vatl_deconv4x4s2_winograd_fwd(const float* x, const float* u, const float* scale, const float* bias, float* y, int N, int H, int W,
                                  int Cin, int Cout, int relu, void* stream);
int64_t vatl_winograd_deconv_stats_row_blocks(int64_t N, int H, int W);
int vatl_deconv4x4s2_winograd_fwd_stats(const float* x, const float* u, float* y, double* stats, int64_t* row_blocks_used, int N, int H, int W,
                                        int Cin, int Cout, void* stream);
/* Weight gradients on the Winograd route (csrc/winograd_wgrad.hip): dw (Cout, Cin, 3, 3) of a 3x3 / stride 1 / pad 1 conv from its input
 * x (N,H,W,Cin) and output gradient dz (N,H,W,Cout); dw (Cin, Cout, 4, 4) of ConvTranspose2d(4,2,1) from its input x (N,H,W,Cin) and output
 * gradient dy (N,2H,2W,Cout).  Replace loss.backward() through those layers (ActiveLearning.py:672) like vatl_conv2d_wgrad /
 * vatl_deconv4x4s2_wgrad; channel counts % 4 == 0, N * tiles < 2^20.  workspace: the *_workspace_floats query; partial sums per tile
 * range are added in a fixed order (bitwise reproducible). */
int64_t vatl_conv3x3_winograd_wgrad_workspace_floats(int Cout, int Cin, int64_t N, int H, int W);
int vatl_conv3x3_winograd_wgrad(const float* x, const float* dz, float* dw, float* workspace, int N, int H, int W, int Cin, int Cout, void* stream);
int64_t vatl_deconv4x4s2_winograd_wgrad_workspace_floats(int Cin, int Cout, int64_t N, int H, int W);
int vatl_deconv4x4s2_winograd_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int H, int W, int Cin, int Cout,
                                    void* stream);
/* Data gradient of ConvTranspose2d(4,2,1) on the Winograd route: dx[ci][y][x] = sum dz[co][2y-1+ky][2x-1+kx] W[ci][co][ky][kx] is a 4x4 /
 * stride 2 conv = the sum over the four pixel phases of dz of 2x2 convolutions: F(3x3,2x2) with the reduction over (phase, channel).
 * w = the layer's (Cin, Cout, 4, 4) weight, Cout % 16 == 0, Cin % 4 == 0; u: vatl_winograd_deconv_dgrad_weight_floats(Cin, Cout) floats.
 * dz (N, 2H, 2W, Cout) -> dx (N, H, W, Cin) (+ residual); _bnbwd: vatl_conv2d_fwd_ex_bnbwd semantics for the consumer layer's BatchNorm,
 * statistics capacity ceil(N * ceil(H/3) * ceil(W/3) / 32) * Cin * 2 doubles. */
int64_t vatl_winograd_deconv_dgrad_weight_floats(int Cin, int Cout);
int vatl_pack_winograd_deconv_dgrad_weight(const float* w, float* u, int Cin, int Cout, void* stream);
int vatl_deconv4x4s2_winograd_dgrad(const float* dz, const float* u, const float* residual, float* dx, int N, int H, int W, int Cin, int Cout,
                                    void* stream);
int vatl_deconv4x4s2_winograd_dgrad_bnbwd(const float* dz, const float* u, const float* residual, float* dx, int N, int H, int W, int Cin,
                                          int Cout, const float* bn_z, const float* bn_mask_y, const float* bn_scale, const float* bn_bias,
                                          const float* bn_mean, const float* bn_invstd, double* stats, int64_t* row_blocks_used, void* stream);
int vatl_conv3x3_winograd_fwd_bnbwd(const float* x, const float* u, const float* residual, float* y, int N, int H, int W, int Cin, int Cout,
                                    const float* bn_z, const float* bn_mask_y, const float* bn_scale, const float* bn_bias,
                                    const float* bn_mean, const float* bn_invstd, double* stats, int64_t* row_blocks_used, void* stream);

/* Parameter guard of the inference plans (csrc/checksum.hip): one launch folds n_tensors device tensors into one 64-bit checksum each.
 * A plan (packed filters, folded BatchNorm) is keyed on tensor addresses and torch version counters; an in-place write through `.data`
 * (`p.data.copy_(w)`: alphapose/models/layers/dcn/deform_conv.py:232,255; hand-written loaders beside ActiveLearning.py:217's
 * load_state_dict) bumps no counter — the checksums taken in front of every plan call, compared with those taken when the plan was
 * built, are what notices it.  table_dev: n_tensors rows of {pointer, 32-bit words, first block} (int64, resident on the device),
 * first block = the running sum of ceil(words / vatl_checksum_block_words()) over the preceding rows, total_blocks = that sum over all
 * rows.  out: n_tensors uint64 (zeroed and written by the call): sum_i (w_i + 1) * (0x9E3779B97F4A7C15 + 2 i) mod 2^64
 * over the tensor's words w_i (every word has its own odd multiplier: any single changed word, and any two swapped words, change the sum) — independent of the order in which blocks finish. */
int64_t vatl_checksum_block_words(void);
int vatl_checksum_multi(const int64_t* table_dev, int n_tensors, int64_t total_blocks, uint64_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VATL_HIP_H */
