/* edrl_hip.h — C-ABI of the MI355X-native EDRL hot path (libedrl_hip.so, gfx950).
 *
 * The reference (xinkunwang111/Robust-Multimodal-Learning-...-Disentangled-Representation) has no
 * FFI/plugin layer: its hot path is `fusion_net.MedFusion.forward` + `MMD.MK_MMD` made of stock
 * torch ops.  Each entry point below replaces one torch op site on that path;
 * the reference site is cited per function (paths relative to the reference repo).
 *
 * Conventions
 *  - plain device pointers and sizes only; no torch types; fp32 data; NHWC activations
 *    ([rows][channels] with an explicit row/pixel stride `ld*` in elements where given).
 *  - every function only enqueues work on `stream` and returns immediately:
 *    0 = ok, >0 = hipError_t, <0 = argument error (-22 EINVAL, -28 workspace too small).
 *  - kernels never allocate; outputs and workspaces are caller-owned (the Python host side
 *    allocates them from the torch caching allocator).
 *  - device scalars (losses and their upstream gradients) are passed as 1-element device
 *    arrays so that nothing on this path synchronises with the host.
 */
#ifndef EDRL_HIP_H
#define EDRL_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

/* flags of edrl_conv2d_nhwc_fwd_f32 / _dgrad_f32 */
#define EDRL_FLAG_RELU 1   /* y = max(y, 0) after bias                     */
#define EDRL_FLAG_ACCUM 2  /* y += previous contents of the destination     */

/* ---- MFMA contractions (conv_gemm.hip) -------------------------------------------------
 * Convolution forward; a Linear layer is the 1x1 case (N = rows, H = W = 1).
 *   y = (relu?)(conv(x, w) + bias) * mul   [+ y]
 * Replaces: encoder convs behind fusion_net.py:884-885 (absent Models/), nn.Linear at
 * fusion_net.py:82-90 (EPRL.encoder, with the ReLU and the Dropout mask fused),
 * :635-643 (DILR projectors), :555 (MHA in/out projections), :562-566 (FFN), :801-805
 * (fc_fundus, fc), and the Gram matmul of code/MMD.py:26.
 * x [N,Hi,Wi,Ci] (pixel stride ld_x), w [Co,KH,KW,Ci], bias [Co]|NULL,
 * mul [N,Ho,Wo,Co]|NULL (pixel stride ld_aux), y [N,Ho,Wo,Co] (pixel stride ld_y). */
int edrl_conv2d_nhwc_fwd_f32(const float* x, const float* w, const float* bias, const float* mul,
                             float* y, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                             int KW, int stride, int pad, long ld_x, long ld_y, long ld_aux, int flags,
                             hipStream_t stream);

/* Same convolution (no bias / activation) with the BatchNorm statistics of its output fused into the epilogue:
 * stat_part [edrl_conv_stats_chunks(N,Ho,Wo)][3][Co] receives per-128-row chunk shifted moments (sum (y-K), sum (y-K)^2, K
 * with K = the chunk's first row), to be reduced by edrl_bn_finalize_partials_f32(rows_per_chunk = 128).
 * stat_shift is reserved (may be NULL).  Needs Ci % 16 == 0. */
long edrl_conv_stats_chunks(int N, int Ho, int Wo);
int edrl_conv2d_nhwc_fwd_stats_f32(const float* x, const float* w, float* y, const float* stat_shift, float* stat_part,
                                   size_t stat_part_bytes, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                                   int KH, int KW, int stride, int pad, hipStream_t stream);
/* The same contraction (fp32 operands, fp32 MFMA, fp32 chunk partials from the accumulators) with y stored as bf16: the stem of
 * the bf16 trunk (fp32 image in, bf16 raw tensor out; replaces the reference trunk's first nn.Conv2d under bf16 storage, SURVEY.md
 * section 8a rows E1/E2).  Ho / Wo are the caller's (the space-to-depth stem pads bottom / right by one less).  Ci % 4 == 0,
 * Co % 4 == 0. */
int edrl_conv2d_nhwc_fwd_stats_f32_obf16(const float* x, const float* w, void* y_bf16, float* stat_part, size_t stat_part_bytes,
                                         int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride,
                                         int pad, hipStream_t stream);

/* The same stem for the 1-channel (OCT) bf16 trunk on the bf16 matrix pipe: xs = the 2x2 space-to-depth image fp32 [N,Hs,Ws,4]
 * (edrl_space_to_depth2_f32), w = the folded 4x4x4 weights as bf16 [64][64] (edrl_stem_weight_fold_f32 + edrl_cast_f32_to_bf16);
 * image and weights are rounded to bf16 in registers, fp32 accumulate, y bf16 [N,Hs,Ws,64], BatchNorm chunk partials
 * [ceil(N*Hs*Ws/128)][3][64] from the fp32 accumulators (stat_part may be NULL).  Streaming kernel: 0.4 GB in, 3.3 GB out per
 * 2048 224x224 slices (fusion_net.py:885 encoder slot, config C2/C4). */
int edrl_stem_conv_s2d_bf16(const float* xs, const void* w_bf16, void* y_bf16, float* stat_part, size_t stat_part_bytes, int N, int Hs,
                            int Ws, hipStream_t stream);

/* Backward of an expanding 1x1 layer (Ci = 64 -> Co = 256, stride 1) inside a fused-BatchNorm block, both gradients from ONE pass
 * over the two Co-wide tensors (the layer1.*.conv3 call sites of the bf16 trunk; replaces one edrl_conv2d_nhwc_wgrad_bn_bf16 + one
 * edrl_conv2d_nhwc_dgrad_bn_bf16 call): d_raw = A*g + nK2*yraw + C2 (bcoef [4][Co]), X = relu(bn(x2raw; x2_fcoef [5][Ci])),
 * dw [Co][Ci] fp32 = d_raw^T X (ordered split-K through `workspace`), g2 [N,H,W,Ci] bf16 = relu'(bn(x2raw)) * (d_raw wt^T) with
 * wt = the permuted weights [Ci][Co] bf16, and the BatchNorm-backward partial sums (sum g2, sum g2*(x2raw - mean)) ->
 * ep_part [edrl_conv1x1_k64_bwd_chunks][2][Ci] (planes = 2 for edrl_bn_bwd_finalize_partials_f32). */
int edrl_conv1x1_k64_bwd_ok_bf16(int N, int H, int W, int Ci, int Co);
long edrl_conv1x1_k64_bwd_chunks(int N, int H, int W);
size_t edrl_conv1x1_k64_bwd_workspace_bytes(int N, int H, int W);
int edrl_conv1x1_k64_bwd_bf16(const void* g, const void* yraw, const float* bcoef, const void* x2raw, const float* x2_fcoef,
                              const void* wt, void* g2, float* ep_part, size_t ep_part_bytes, float* dw, float* workspace,
                              size_t workspace_bytes, int N, int H, int W, int Ci, int Co, hipStream_t stream);

/* Weight gradient of that stem on the bf16 matrix pipe: dy = d_raw of the stem's BatchNorm, bf16 [N,Hs,Ws,64]; xs as above; the image
 * is rounded to bf16 in registers; dw fp32 [64][4][4][4] (folded layout; edrl_stem_weight_fold_f32 dir 1 gathers the 49 taps),
 * ordered split reduction through `workspace` (edrl_stem_wgrad_s2d_bf16_workspace_bytes). */
size_t edrl_stem_wgrad_s2d_bf16_workspace_bytes(int N, int Hs, int Ws);
int edrl_stem_wgrad_s2d_bf16(const void* dy, const float* xs, float* dw, float* workspace, size_t workspace_bytes, int N, int Hs, int Ws,
                             hipStream_t stream);

/* Data gradient (autograd of the above): dx [+]= conv_transpose(dy, w).
 * wt is w permuted to [Ci,KH,KW,Co] by edrl_permute_weight_f32. */
int edrl_conv2d_nhwc_dgrad_f32(const float* dy, const float* wt, float* dx, int N, int Hi, int Wi, int Ci,
                               int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, long ld_dy,
                               long ld_dx, int flags, hipStream_t stream);

/* Weight gradient: dw[Co,KH,KW,Ci] [+]= sum over pixels dy (x) x, split-K with an ordered
 * (deterministic) reduction through `workspace`. */
size_t edrl_conv2d_nhwc_wgrad_workspace_bytes(int N, int Ho, int Wo, int Co, int Ci, int KH, int KW);
int edrl_conv2d_nhwc_wgrad_f32(const float* dy, const float* x, float* dw, float* workspace,
                               size_t workspace_bytes, int N, int Hi, int Wi, int Ci, int Ho, int Wo,
                               int Co, int KH, int KW, int stride, int pad, long ld_dy, long ld_x,
                               int accumulate, hipStream_t stream);
/* The same weight gradient with dy stored as bf16 (dense [N,Ho,Wo,Co], widened exactly on load) and x fp32: the stem of the bf16
 * trunk.  Co <= 64, Co % 4 == 0, Ci % 4 == 0; workspace as edrl_conv2d_nhwc_wgrad_workspace_bytes. */
int edrl_conv2d_nhwc_wgrad_f32_dybf16_ok(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW);  /* 1 / 0 */
int edrl_conv2d_nhwc_wgrad_f32_dybf16(const void* dy_bf16, const float* x, float* dw, float* workspace, size_t workspace_bytes,
                                      int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad,
                                      int accumulate, hipStream_t stream);

/* ---- Convolutions with the neighbouring BatchNorm passes folded in (encoder slots behind fusion_net.py:884-885; SURVEY.md
 * §8a rows E1-E3: conv -> BatchNorm2d(train) -> ReLU chains).  Dense NHWC fp32 tensors, 16-byte aligned.  Coefficient arrays:
 *   fcoef [5][C] = {mean, rstd, scale = gamma*rstd, shift = beta, shift2 = shift - mean*scale}   (edrl_bn_finalize_fcoef_f32)
 *   bcoef [4][C] = {A = gamma*rstd, nK2 = -A*rstd*mean(g*xhat), C2 = -nK2*mean - A*mean(g), mean}
 *                                                                                    (edrl_bn_bwd_finalize_partials_f32)
 * so that relu(x*scale + shift2) and d_raw = A*g + nK2*x + C2 are formed inside the conv kernels' operand loads and
 * the activated tensors / d_raw tensors (torch: F.batch_norm + F.relu outputs and their autograd buffers) never exist.
 * edrl_conv2d_fused_ok_f32 says whether a layer geometry has these paths (1) or must use the separate passes (0). */
int edrl_conv2d_fused_ok_f32(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad);
/* y = conv(relu(bn(x_raw; in_fcoef)), w), BatchNorm chunk partials of y -> stat_part (as edrl_conv2d_nhwc_fwd_stats_f32). */
int edrl_conv2d_nhwc_fwd_bnin_stats_f32(const float* x, const float* in_fcoef, const float* w, float* y, float* stat_part,
                                        size_t stat_part_bytes, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                                        int KW, int stride, int pad, hipStream_t stream);
/* dx [+]= conv_transpose(d_raw(g, yraw; bcoef), w).  With ep_raw != NULL, dx is the gradient of relu?(bn(ep_raw)) of the layer
 * below: the epilogue masks it (ep_mask sign bytes [pixel][Ci/4], or recomputed from ep_raw / ep_fcoef when ep_mask == NULL and
 * ep_relu), stores the masked gradient and writes (sum g, sum g*(x - mean)), mean = ep_fcoef row 0, per 128-row tile to
 * ep_part [edrl_conv_dgrad_bn_chunks(N,Hi,Wi,stride,pad)][2][Ci].  flags: 2 = accumulate into dx before masking. */
long edrl_conv_dgrad_bn_chunks(int N, int Hi, int Wi, int stride, int pad);
int edrl_conv2d_nhwc_dgrad_bn_f32(const float* g, const float* yraw, const float* bcoef, const float* wt, float* dx, int N,
                                  int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad,
                                  int flags, const float* ep_raw, const unsigned char* ep_mask, const float* ep_fcoef,
                                  int ep_relu, float* ep_part, size_t ep_part_bytes, hipStream_t stream);
/* dw [+]= sum over pixels d_raw(g, yraw; bcoef) (x) X, X = relu(bn(x; x_fcoef)) or x itself (x_fcoef == NULL). */
int edrl_conv2d_nhwc_wgrad_bn_f32(const float* g, const float* yraw, const float* bcoef, const float* x,
                                  const float* x_fcoef, float* dw, float* workspace, size_t workspace_bytes, int N, int Hi,
                                  int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, int accumulate,
                                  hipStream_t stream);

/* in [A][B][C] -> out [C][B][A]. */
int edrl_permute_weight_f32(const float* in, float* out, int A, int B, int C, hipStream_t stream);

/* ---- bf16 contractions (conv_bf16.hip): the C2/C4 precision of SURVEY.md §8(a) rows E1/E2 ------------------------
 * bf16 operands (device pointers to IEEE bfloat16), fp32 accumulate on v_mfma_f32_32x32x16_bf16, bf16 results.
 * Forward: Ci % 32 == 0, Co % 4 == 0; stat_part (optional) receives the fp32 BatchNorm chunk partials of the
 * accumulators exactly as edrl_conv2d_nhwc_fwd_stats_f32.  Data gradient: Co % 32 == 0, Ci % 4 == 0, wt = bf16
 * [Ci,KH,KW,Co] (edrl_permute_weight_bf16), flags = EDRL_FLAG_ACCUM or 0. */
int edrl_conv2d_nhwc_fwd_bf16(const void* x, const void* w, void* y, float* stat_part, size_t stat_part_bytes, int N,
                              int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad,
                              hipStream_t stream);
int edrl_conv2d_nhwc_dgrad_bf16(const void* dy, const void* wt, void* dx, int N, int Hi, int Wi, int Ci, int Ho, int Wo,
                                int Co, int KH, int KW, int stride, int pad, int flags, hipStream_t stream);
/* Weight gradient: dw fp32 [Co,KH,KW,Ci] [+]= sum_pix dy(bf16) (x) x(bf16); transposing LDS reads (ds_read_b64_tr_b16),
 * split-K with ordered reduction.  Co % 8 == 0, Ci % 8 == 0. */
size_t edrl_conv2d_nhwc_wgrad_bf16_workspace_bytes(int N, int Ho, int Wo, int Co, int Ci, int KH, int KW);
int edrl_conv2d_nhwc_wgrad_bf16(const void* dy, const void* x, float* dw, float* workspace, size_t workspace_bytes, int N,
                                int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad,
                                int accumulate, hipStream_t stream);
int edrl_cast_f32_to_bf16(const float* in, void* out, long n, hipStream_t stream);
int edrl_cast_bf16_to_f32(const void* in, float* out, long n, hipStream_t stream);
int edrl_permute_weight_bf16(const float* in, void* out, int A, int B, int C, hipStream_t stream);

/* ---- BatchNorm / pooling / layout (bn_pool.hip) -----------------------------------------
 * Train-mode BatchNorm over rows of x [M][C]: batch statistics, running-stat update
 * (momentum, unbiased variance), and the per-channel affine (scale, shift) that applies it.
 * Replaces F.batch_norm(training=True) inside the encoders and DILR.bn1/bn2
 * (fusion_net.py:653-654,658,757-758; gamma = beta = NULL for affine=False). */
size_t edrl_bn_workspace_bytes(long M, int C);
int edrl_bn_train_stats_f32(const float* x, long M, int C, long ld, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps,
                            float* save_mean, float* save_rstd, float* scale, float* shift, float* workspace,
                            size_t workspace_bytes, hipStream_t stream);
/* Reduce chunk partials [nchunks][3][C] (shifted moments; chunk k = rows [k*rows_per_chunk, ...)) to the batch statistics and
 * the BN affine, exactly as the second half of edrl_bn_train_stats_f32.  group_ws (optional, fp64,
 * edrl_bn_finalize_group_ws_bytes) enables the two-stage reduction used for large chunk counts. */
size_t edrl_bn_finalize_group_ws_bytes(long nchunks, int C);
int edrl_bn_finalize_partials_f32(const float* part, long nchunks, int rows_per_chunk, long M, int C, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                  float* save_mean, float* save_rstd, float* scale, float* shift, double* group_ws,
                                  size_t group_ws_bytes, hipStream_t stream);
/* out = (relu?)((x - mean)*scale + shift [+ residual])   (scale = gamma*rstd, shift = beta).
 * relu_mask (optional, dense [M][C/4] bytes, needs ld == C): 4 ReLU sign bits per 4 channels, so the backward
 * reads 1 byte instead of the 16-byte activation. */
int edrl_bn_apply_f32(const float* x, const float* mean, const float* scale, const float* shift,
                      const float* residual, float* out, unsigned char* relu_mask, long M, int C, long ld, int relu,
                      hipStream_t stream);
/* Backward of BN(+residual)(+ReLU).  dout = grad of the activated output; the ReLU mask comes from relu_mask
 * (preferred) or from out > 0 (both NULL = no ReLU); dx = grad of the raw input; dres (optional) [+]= masked dout.
 * workspace >= edrl_bn_workspace_bytes(M,C) + 2*C*4 bytes. */
int edrl_bn_bwd_f32(const float* dout, const float* out, const unsigned char* relu_mask, const float* x,
                    const float* save_mean, const float* save_rstd, const float* gamma, float* dgamma, float* dbeta,
                    int accumulate, float* dx, float* dres, int dres_accum, long M, int C, long ld, float* workspace,
                    size_t workspace_bytes, hipStream_t stream);

/* edrl_bn_finalize_partials_f32 with the result as ONE array fcoef [5][C] = {mean, rstd, scale, shift, shift2 = shift -
 * mean*scale} (the form the fused conv kernels take: relu(x*scale + shift2) is one packed FMA + max per element). */
int edrl_bn_finalize_fcoef_f32(const float* part, long nchunks, int rows_per_chunk, long M, int C, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* fcoef, double* group_ws, size_t group_ws_bytes, hipStream_t stream);
/* BatchNorm apply whose residual operand may be a raw conv output with its own BatchNorm (downsample branch; res_fcoef NULL:
 * plain residual): out = act(bn(x; fcoef) + bn(residual; res_fcoef)), ReLU sign bytes -> relu_mask (optional). */
int edrl_bn_apply_res_f32(const float* x, const float* fcoef, const float* residual, const float* res_fcoef, float* out,
                          unsigned char* relu_mask, long M, int C, int relu, hipStream_t stream);
/* BatchNorm(+ReLU) backward split for the fused conv kernels.  _reduce: g = dout * relu-mask (optional) -> g_out (optional),
 * partial sums (sum g, sum g*xhat) -> part [ceil(M/1024)][3][C] (edrl_bn_workspace_bytes).  _finalize: partial sums with
 * `planes` planes per chunk (3 from _reduce; 2 from edrl_conv2d_nhwc_dgrad_bn_f32, whose second plane is sum g*(x - mean) and is
 * scaled by rstd here in fp64) -> dgamma, dbeta, bcoef [4][C]. */
int edrl_bn_bwd_reduce_f32(const float* dout, const unsigned char* relu_mask, const float* x, const float* fcoef, float* g_out,
                           float* part, size_t part_bytes, long M, int C, hipStream_t stream);
size_t edrl_bn_bwd_group_ws_bytes(long nchunks, int C);
int edrl_bn_bwd_finalize_partials_f32(const float* part, long nchunks, int planes, long M, int C, const float* gamma,
                                      const float* fcoef, float* dgamma, float* dbeta, float* bcoef, double* group_ws,
                                      size_t group_ws_bytes, hipStream_t stream);

/* Stem (conv 7x7/s2 -> BatchNorm -> ReLU -> max-pool 3x3/s2/p1, the head of the encoder slots behind fusion_net.py:884-885) with
 * the BatchNorm + ReLU folded into the max-pool: the activated stem tensor and its sign bytes are never stored.
 * x = RAW stem conv output [N,H,W,C]; fcoef [5][C] from edrl_bn_train_stats_fcoef_f32; backward = _reduce (partial sums, planes = 3
 * for edrl_bn_bwd_finalize_partials_f32) then _apply (d_raw = A*g + nK2*x + C2 for the stem weight gradient). */
int edrl_bn_train_stats_fcoef_f32(const float* x, long M, int C, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float momentum, float eps, float* fcoef, float* workspace,
                                  size_t workspace_bytes, hipStream_t stream);
int edrl_maxpool3x3s2_bn_fwd_f32(const float* x, const float* fcoef, float* y, unsigned char* idx, int N, int H, int W, int C,
                                 hipStream_t stream);
int edrl_maxpool3x3s2_bn_bwd_reduce_f32(const float* dy, const unsigned char* idx, const float* x, const float* fcoef, float* part,
                                        size_t part_bytes, int N, int H, int W, int C, hipStream_t stream);
int edrl_maxpool3x3s2_bn_bwd_apply_f32(const float* dy, const unsigned char* idx, const float* x, const float* fcoef,
                                       const float* bcoef, float* d_raw, int N, int H, int W, int C, hipStream_t stream);
/* The same three with the pooled tensor (y) / its gradient (dy) stored as bf16 when y_bf16 / dy_bf16 is 1, and the raw stem conv
 * output x stored as bf16 when x_bf16 is 1 (needs the other flag too): the stem of the bf16 trunk (C2/C4; the statistics and
 * d_raw is fp32, or bf16 with d_bf16 = 1 (needs x_bf16; consumed by edrl_conv2d_nhwc_wgrad_f32_dybf16)). */
int edrl_maxpool3x3s2_bn_fwd_mx(const void* x, int x_bf16, const float* fcoef, void* y, int y_bf16, unsigned char* idx, int N, int H,
                                int W, int C, hipStream_t stream);
int edrl_maxpool3x3s2_bn_bwd_reduce_mx(const void* dy, int dy_bf16, const unsigned char* idx, const void* x, int x_bf16,
                                       const float* fcoef, float* part, size_t part_bytes, int N, int H, int W, int C,
                                       hipStream_t stream);
int edrl_maxpool3x3s2_bn_bwd_apply_mx(const void* dy, int dy_bf16, const unsigned char* idx, const void* x, int x_bf16,
                                      const float* fcoef, const float* bcoef, void* d_raw, int d_bf16, int N, int H, int W, int C,
                                      hipStream_t stream);

/* bf16 counterparts of the fused-BatchNorm entry points (conv_bf16.hip / bn_pool.hip): bf16 tensors, fp32 coefficient arrays
 * fcoef [5][C] / bcoef [4][C] and fp32 partial sums; same contracts as the _f32 versions above.  Ci % 32 == 0, Co % 32 == 0. */
int edrl_conv2d_fused_ok_bf16(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad);
int edrl_conv2d_nhwc_fwd_bnin_stats_bf16(const void* x, const float* in_fcoef, const void* w, void* y, float* stat_part,
                                         size_t stat_part_bytes, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                                         int KW, int stride, int pad, hipStream_t stream);
int edrl_conv2d_nhwc_dgrad_bn_bf16(const void* g, const void* yraw, const float* bcoef, const void* wt, void* dx, int N,
                                   int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad,
                                   int flags, const void* ep_raw, const unsigned char* ep_mask, const float* ep_fcoef,
                                   int ep_relu, float* ep_part, size_t ep_part_bytes, hipStream_t stream);
int edrl_conv2d_nhwc_wgrad_bn_bf16(const void* g, const void* yraw, const float* bcoef, const void* x, const float* x_fcoef,
                                   float* dw, float* workspace, size_t workspace_bytes, int N, int Hi, int Wi, int Ci, int Ho,
                                   int Wo, int Co, int KH, int KW, int stride, int pad, int accumulate, hipStream_t stream);
int edrl_bn_apply_res_bf16(const void* x, const float* fcoef, const void* residual, const float* res_fcoef, void* out,
                           unsigned char* relu_mask, long M, int C, int relu, hipStream_t stream);
int edrl_bn_bwd_reduce_bf16(const void* dout, const unsigned char* relu_mask, const void* x, const float* fcoef, void* g_out,
                            float* part, size_t part_bytes, long M, int C, hipStream_t stream);
/* d_raw (bf16) = A*g + nK2*x + C2 with bcoef [4][C]: the BatchNorm-backward apply step as a standalone pass, for a conv that takes
 * d_raw on the plain kernels inside an otherwise fused block (the 3x3 layer of a bf16 bottleneck block). */
int edrl_bn_draw_bf16(const void* g, const void* x, const float* bcoef, void* d_raw, long M, int C, hipStream_t stream);
/* The same pass over fp32 tensors (C % 4 == 0): the fp32 trunk's units that run on the plain kernels (encoders._K32: the 3x3
 * layer of a fused bottleneck block, the blocks above EDRL_F32_FUSE_MAXPLANES). */
int edrl_bn_draw_f32(const float* g, const float* x, const float* bcoef, float* d_raw, long M, int C, hipStream_t stream);

/* Mixed-precision BatchNorm apply / backward and max-pool of the bf16 (C2) trunk: raw_bf16 / act_bf16 give the storage type
 * (0 fp32, 1 bf16) of the raw conv output (+ its gradient) and of the activated tensors (+ their gradients); statistics,
 * affine and reductions stay fp32/fp64.  Dense rows. */
int edrl_bn_apply_mx(const void* x, int raw_bf16, const float* mean, const float* scale, const float* shift,
                     const void* residual, void* out, int act_bf16, unsigned char* relu_mask, long M, int C, int relu,
                     hipStream_t stream);
int edrl_bn_bwd_mx(const void* dout, int act_bf16, const unsigned char* relu_mask, const void* x, int raw_bf16,
                   const float* save_mean, const float* save_rstd, const float* gamma, float* dgamma, float* dbeta, void* dx,
                   void* dres, long M, int C, float* workspace, size_t workspace_bytes, hipStream_t stream);
int edrl_maxpool3x3s2_fwd_bf16(const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, hipStream_t stream);
int edrl_maxpool3x3s2_bwd_bf16(const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C,
                               hipStream_t stream);
/* 3x3 / stride 2 / pad 1 max pooling on NHWC (encoder stem); idx = window tap of the first max. */
/* Stem re-layout (build-owned encoders): the 7x7/s2/p3 stem conv == a 4x4/s1 conv on the 2x2
 * space-to-depth image (pads 2 top/left, 1 bottom/right).  y[n,a,b,(ph*2+pw)*C+c] = x[n,2a+ph,2b+pw,c]; H, W even. */
int edrl_space_to_depth2_f32(const float* x, float* y, int N, int H, int W, int C, hipStream_t stream);
/* dir 0: w [Co,7,7,C] -> [Co,4,4,4C] (the unused 8th taps zero);  dir 1: the inverse gather (for the weight gradient). */
int edrl_stem_weight_fold_f32(const float* in, float* out, int Co, int C, int dir, hipStream_t stream);
int edrl_maxpool3x3s2_fwd_f32(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C,
                              hipStream_t stream);
int edrl_maxpool3x3s2_bwd_f32(const float* dy, const unsigned char* idx, float* dx, int N, int H, int W, int C,
                              hipStream_t stream);
/* [N,C,H,W] -> [N,H,W,Cp], channels C..Cp-1 zero (loader layout of data_harvard.py:830-841). */
int edrl_nchw_to_nhwc_f32(const float* in, float* out, int N, int C, int H, int W, int Cp, hipStream_t stream);
/* out[a][d] = scale * sum_l in[a][l][d]  — global average pool; att.mean(dim=1) fusion_net.py:225;
 * torch.mean(y_uni, dim=1) fusion_net.py:737-738; bias gradients. */
int edrl_sum_axis1_f32(const float* in, float* out, long A, int L, int D, float scale, hipStream_t stream);
/* out[a][l][d] [+]= scale * in[a][d]  — its backward; mu_proxy.repeat fusion_net.py:246-247. */
int edrl_bcast_axis1_f32(const float* in, float* out, long A, int L, int D, float scale, int accumulate,
                         hipStream_t stream);

/* ---- head reductions (head_ops.hip) ---------------------------------------------------- */
/* op codes of edrl_ew_f32 */
#define EDRL_EW_RELU 0
#define EDRL_EW_RELU_BWD 1
#define EDRL_EW_AXPBY 2
#define EDRL_EW_MUL 3
#define EDRL_EW_SCALE 4
#define EDRL_EW_SOFTPLUS 5
#define EDRL_EW_SOFTPLUS_BWD 6
#define EDRL_EW_MASKED_BWD 7
#define EDRL_EW_ADD_RELU 8
#define EDRL_EW_SCALE_BY_PTR 9
#define EDRL_EW_FILL 10
#define EDRL_EW_LERP_BY_PTR 11
int edrl_ew_f32(int op, long n, const float* a, const float* b, const float* c, float* out, float alpha,
                float beta, hipStream_t stream);
/* out = sum_i w[i] * *in[i]  (n <= 8) — loss mixers fusion_net.py:870-879,942-948, fusion_train.py:212.
 * `in` and `w` are HOST arrays (of device pointers / of weights). */
int edrl_scalar_mix_f32(const float* const* in, const float* w, int n, float* out, hipStream_t stream);

/* F.normalize over axis 1 of [A][L][D] (fusion_net.py:149-150). inv [A][D] saved for backward. */
int edrl_l2norm_axis1_fwd_f32(const float* x, float* y, float* inv, long A, int L, int D, float eps,
                              hipStream_t stream);
int edrl_l2norm_axis1_bwd_f32(const float* dy, const float* y, const float* inv, float* dx, long A, int L, int D,
                              hipStream_t stream);
/* out[a][l][d] = u[a][d] + v[a][d]*w[a][l][d] (fusion_net.py:143-146, 907, 910) and its backward. */
int edrl_affine_bcast_fwd_f32(const float* u, const float* v, const float* w, float* out, long A, int L, int D,
                              hipStream_t stream);
int edrl_affine_bcast_bwd_f32(const float* dout, const float* w, float* du, float* dv, long A, int L, int D,
                              hipStream_t stream);

/* Essence-point selection (fusion_net.py:227-243): positive = row of the label's proxy,
 * negative = the remaining rows; top-K of each, loss = mean_b exp(neg_mean - pos_mean).
 * att [B][C][S]; y int64 [B]; sel [B][C][S] (caller zeroes) marks the selected entries
 * (bit-exact index set); means [B][2]; e [B]; loss [1]. */
int edrl_topk_margin_fwd_f32(const float* att, const long long* y, unsigned char* sel, float* means, float* e,
                             float* loss, int B, int C, int S, int K, hipStream_t stream);
int edrl_topk_margin_bwd_f32(const float* dloss, const float* e, const unsigned char* sel, const long long* y,
                             float* datt, int B, int C, int S, int K, hipStream_t stream);
/* flag[0] |= 1 when a label is outside [0,C) (the reference raises KeyError, fusion_net.py:101,227). */
int edrl_check_labels(const long long* y, int B, int C, int* flag, hipStream_t stream);

/* PoE.forward for two experts (fusion_net.py:26-52), elementwise over R; workspace >= 128 floats. */
int edrl_poe2_fwd_f32(const float* mu0, const float* s0, const float* mu1, const float* s1, const float* phi,
                      float* out, long R, float eps, hipStream_t stream);
int edrl_poe2_bwd_f32(const float* g, const float* mu0, const float* s0, const float* mu1, const float* s1,
                      const float* phi, float* dmu0, float* ds0, float* dmu1, float* ds1, float* dphi,
                      float* workspace, long R, float eps, hipStream_t stream);
/* get_KL_loss / KL_between_normals vs N(0,1) (fusion_net.py:390-402,838-850); mu, sg [Bn][C][D];
 * workspace >= 64 floats. */
int edrl_kl_normal_fwd_f32(const float* mu, const float* sg, float* loss, float* workspace, long Bn, int C, int D,
                           hipStream_t stream);
int edrl_kl_normal_bwd_f32(const float* dloss, const float* mu, const float* sg, float* dmu, float* dsg, long Bn,
                           int C, int D, hipStream_t stream);

/* Inner attention of nn.MultiheadAttention(E, H) with head dim 128 (fusion_net.py:555,571):
 * q [B][Lq][E] (projected), kv [B][N][2E] (projected keys | values), P [B][H][Lq][N], ctx [B][Lq][E]. */
int edrl_mha_core_fwd_f32(const float* q, const float* kv, float* P, float* ctx, int B, int Lq, int N, int H, int E,
                          hipStream_t stream);
int edrl_mha_core_bwd_f32(const float* dctx, const float* q, const float* kv, const float* P, float* dq, float* dkv,
                          int B, int Lq, int N, int H, int E, hipStream_t stream);

/* nn.LayerNorm(E) over rows (fusion_net.py:560,573). */
int edrl_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                           long R, int E, float eps, hipStream_t stream);
int edrl_layernorm_bwd_f32(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                           float* dx, float* dw, float* db, long R, int E, hipStream_t stream);

/* DILR.bt_loss_cross reductions (fusion_net.py:664-677) on the two diagonal blocks of c.
 * out[7] = loss_c,on_c,off_c,loss_u,on_u,off_u,(loss_c+loss_u)/2; workspace >= 512 floats. */
int edrl_bt_loss_fwd_f32(const float* cc, const float* cu, int n, float lambd, float* out, float* workspace,
                         hipStream_t stream);
int edrl_bt_loss_bwd_f32(const float* dloss12, const float* cc, const float* cu, float* dcc, float* dcu, int n,
                         float lambd, hipStream_t stream);

/* Label-smoothed cross entropy (fusion_net.py:931-939) and argmax (fusion_train.py:213). */
int edrl_smooth_ce_fwd_f32(const float* pred, const long long* y, float* loss, int B, int C, float smoothing,
                           hipStream_t stream);
int edrl_smooth_ce_bwd_f32(const float* dloss, const float* pred, const long long* y, float* dpred, int B, int C,
                           float smoothing, hipStream_t stream);
int edrl_argmax_rows_f32(const float* x, long long* out, int B, int C, hipStream_t stream);

/* ---- eval branch of EPRL / eval-mode BatchNorm (fusion_net.py:152-218) ---- */
int edrl_softmax_rows_f32(const float* x, float* y, int R, int C, hipStream_t stream);
/* out[r] = scale * sum_d x[r][d]  — torch.mean(att, dim=2), torch.mean(z_norm, dim=2) (fusion_net.py:162-163). */
int edrl_rowsum_f32(const float* x, float* out, long R, int D, long ld, float scale, hipStream_t stream);
/* torch.max(combined, dim=1), confidence > threshold, fall back to the most confident sample
 * (fusion_net.py:177-184); labels int64 [B], keep uint8 [B], count int32 [1]: bit-exact index ops. */
int edrl_pseudo_label_f32(const float* comb, int B, int C, float threshold, long long* labels, unsigned char* keep,
                          int* count, hipStream_t stream);
/* EPRL.entropy_regularization (fusion_net.py:127-131) -> out[1]. */
int edrl_entropy_rows_f32(const float* x, float* out, int R, int C, hipStream_t stream);
/* eval-mode BatchNorm: scale = gamma/sqrt(running_var+eps), shift = beta; apply with edrl_bn_apply_f32(mean=running_mean). */
int edrl_bn_eval_params_f32(const float* gamma, const float* beta, const float* running_var, float eps, float* scale,
                            float* shift, int C, hipStream_t stream);

/* ---- SURVEY §8(f) rows 3-4 ---- */
/* out = clip(x + sigma*noise, 0, 1): the high-noise twin view of data_harvard.py:769-783 made on the device. */
int edrl_twin_view_f32(const float* x, const float* noise, float* out, long n, float sigma, hipStream_t stream);
/* Salt-and-pepper (add_salt_peper / add_salt_peper_3D, data_harvard.py:24-48): x[img,:,rows[img][j],cols[img][j]] = value,
 * x NCHW [n_img,C,H,W] in place, rows/cols int32 [n_img][n_pts] drawn by the caller; out-of-range points are ignored. */
int edrl_scatter_fill_nchw_f32(float* x, const int* rows, const int* cols, int n_img, long n_pts, int C, int H, int W,
                               float value, hipStream_t stream);
/* compute_kl_divergence(p, m) = mean_b sum_c p log(p/m) (code/MMD.py:92-95; compute_js_divergence :76-90 composes it). */
int edrl_kl_rows_fwd_f32(const float* p, const float* m, float* out, int B, int C, hipStream_t stream);
int edrl_kl_rows_bwd_f32(const float* dloss, const float* p, const float* m, float* dp, float* dm, int B, int C,
                         hipStream_t stream);

/* ---- MK-MMD (mmd.hip; code/MMD.py:3-74) ------------------------------------------------- */
int edrl_rowsq_f32(const float* x, float* sq, int n, int d, long ld, hipStream_t stream);
/* G = total@total^T [n][n], sq [n]; loss [1]; saved [3] = {bandwidth, signed sum, loss}. */
int edrl_mk_mmd_fwd_f32(const float* G, const float* sq, int n, int ns, float kernel_mul, int kernel_num, float* loss,
                        float* saved, hipStream_t stream);
/* coef [n][n] with dTotal = coef @ total; workspace n*n floats. */
int edrl_mk_mmd_bwd_f32(const float* dloss, const float* G, const float* sq, const float* saved, int n, int ns,
                        float kernel_mul, int kernel_num, float* workspace, float* coef, hipStream_t stream);

/* ---- volume operators (vol_ops.hip): the 3-D-conv OCT encoder alternative of SURVEY.md §8(f) row 4 (build-owned; the
 * reference's 3-D encoder source is absent, call site fusion_net.py:799,885; shapes after baseline_models.py:154-178) ----
 * Depth unfold: y[n,do,p,kd*C+c] = x[n, do*sd - pd + kd, p, c] (0 outside; channels >= KD*C are zero padding up to CK), so a
 * KD x k x k conv is the 2-D conv of edrl_conv2d_nhwc_* over N*Do slices with CK channels; depth fold is its adjoint. */
int edrl_depth_unfold_f32(const float* x, float* y, int N, int D, long P, int C, int KD, int sd, int pd, int Do, int CK,
                          hipStream_t stream);
int edrl_depth_fold_f32(const float* dy, float* dx, int N, int D, long P, int C, int KD, int sd, int pd, int Do, int CK,
                        hipStream_t stream);
/* The same pair on bf16 tensors (the bf16 3-D trunk, encoders3d.py: its convolutions run on the bf16 MFMA kernels over the
 * depth-unfolded operand): C % 8 == 0, unfolded width exactly KD*C, 16-byte aligned; the fold sums its taps in fp32 and rounds once. */
int edrl_depth_unfold_bf16(const void* x_bf16, void* y_bf16, int N, int D, long P, int C, int KD, int sd, int pd, int Do, hipStream_t stream);
int edrl_depth_fold_bf16(const void* dy_bf16, void* dx_bf16, int N, int D, long P, int C, int KD, int sd, int pd, int Do, hipStream_t stream);
/* Depth half of MaxPool3d(3, stride 2, pad 1) on [N,D,PC] (PC = H*W*C); idx = winning tap 0..2. */
int edrl_maxpool_depth3s2_fwd_f32(const float* x, float* y, unsigned char* idx, int N, int D, long PC, hipStream_t stream);
int edrl_maxpool_depth3s2_bwd_f32(const float* dy, const unsigned char* idx, float* dx, int N, int D, long PC,
                                  hipStream_t stream);

/* ---- optimiser (optim.hip) -------------------------------------------------------------------------------------
 * Fused multi-tensor Adam: replaces `optimizer.step()` of torch.optim.Adam(model.parameters(), lr, weight_decay=1e-6)
 * (fusion_train.py:224, :747).  tensors: device array of {float* p; const float* g; float* m;
 * float* v; long n} records; chunks: device array of {int tensor; int chunk} records, edrl_adam_chunk_elems() elements
 * per chunk; step = the step count after this update (>= 1). */
int edrl_adam_chunk_elems(void);
int edrl_adam_multi_f32(const void* tensors, int n_tensors, const void* chunks, int n_chunks, double lr, double beta1,
                        double beta2, double eps, double weight_decay, long step, hipStream_t stream);

/* Weight shadows of a whole encoder trunk in one launch: for every conv weight w fp32 [A][B][C] (= [Co][KH*KW][Ci]) the bf16
 * forward operand `cast` (same layout; NULL for the fp32 trunk) and the data-gradient operand `perm` [C][B][A] (bf16 when
 * perm_bf16 != 0, else fp32; NULL to skip) -- the values of edrl_cast_f32_to_bf16 / edrl_permute_weight_{f32,bf16}.  The host
 * (encoders.ResNetTrunk.build_shadows, called once per train_step before the first view) replaces one cast + one permute launch per
 * layer and view with this.  tensors: device array of {const float* w; void* cast; void* perm; int A, B, C, perm_bf16; long n}
 * records (48 bytes); chunks as for edrl_adam_multi_f32. */
int edrl_weight_shadows_multi(const void* tensors, int n_tensors, const void* chunks, int n_chunks, hipStream_t stream);

/* 3-D convolution forward over NDHWC volumes without the depth-unfolded copy (the true-3-D OCT encoder of SURVEY.md section 8f row 4;
 * layer shapes of the reference's 3-D networks, baseline_models.py:154-178): x [N,Di,Hi,Wi,Ci], w [Co,KH,KW,KD*Ci] (the
 * depth-unfolded weight layout, depth tap innermost of the taps), y [N,Do,Ho,Wo,Co]; depth stride / padding dstride / dpad,
 * in-plane stride / pad.  Ci % 16 == 0, Co % 4 == 0, 16-byte aligned tensors, per-tile source footprint < 2 GiB:
 * edrl_conv3d_fwd_ok_f32; callers fall back to edrl_depth_unfold_f32 + edrl_conv2d_nhwc_fwd_f32 otherwise (the 1-channel stem). */
int edrl_conv3d_fwd_ok_f32(int N, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co, int KD, int KH, int KW);
int edrl_conv3d_ndhwc_fwd_f32(const float* x, const float* w, float* y, int N, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo,
                              int Co, int KD, int KH, int KW, int dstride, int stride, int dpad, int pad, hipStream_t stream);

/* 3-D convolution data gradient over NDHWC volumes without the k_d-times unfolded gradient and its fold pass (same row of SURVEY.md
 * section 8f; the gradient of the layers above, baseline_models.py:154-178): dy [N,Do,Ho,Wo,Co] -> dx [N,Di,Hi,Wi,Ci].  `wt3` is
 * the class-wise permutation of the forward weight made by edrl_conv3d_dgrad_weight_f32 (Co*KH*KW*KD*Ci floats: per depth tap class
 * kd % dstride the matrix [Ci][KH][KW][taps of the class, reversed][Co]).  Co % 16 == 0, Ci % 4 == 0, Di % dstride == 0,
 * dstride == 1 or dstride == stride, power-of-two stride, 16-byte aligned tensors: edrl_conv3d_dgrad_ok_f32; callers fall back to
 * edrl_conv2d_nhwc_dgrad_f32 on the unfolded form + edrl_depth_fold_f32 otherwise. */
int edrl_conv3d_dgrad_ok_f32(int N, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co, int KD, int KH, int KW,
                             int dstride, int stride);
int edrl_conv3d_dgrad_weight_f32(const float* w, float* wt3, int Co, int KH, int KW, int KD, int Ci, int dstride, hipStream_t stream);
int edrl_conv3d_ndhwc_dgrad_f32(const float* dy, const float* wt3, float* dx, int N, int Di, int Hi, int Wi, int Ci, int Do, int Ho,
                                int Wo, int Co, int KD, int KH, int KW, int dstride, int stride, int dpad, int pad, hipStream_t stream);

/* 3-D convolution weight gradient over NDHWC volumes without the depth-unfolded operand (same row): dy [N,Do,Ho,Wo,Co],
 * x [N,Di,Hi,Wi,Ci] -> dw [Co,KH,KW,KD*Ci] (the depth-unfolded weight layout).  Workspace =
 * edrl_conv3d_wgrad_workspace_bytes(N, Do, Ho, Wo, Co, Ci, KD, KH, KW) (the launcher's own split-K plan).  Ci % 4 == 0, Co % 4 == 0,
 * dense 16-byte aligned tensors,
 * buffer-load path geometry: edrl_conv3d_wgrad_ok_f32; callers fall back to edrl_depth_unfold_f32 + edrl_conv2d_nhwc_wgrad_f32. */
int edrl_conv3d_wgrad_ok_f32(int N, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co, int KD, int KH, int KW);
size_t edrl_conv3d_wgrad_workspace_bytes(int N, int Do, int Ho, int Wo, int Co, int Ci, int KD, int KH, int KW);
int edrl_conv3d_ndhwc_wgrad_f32(const float* dy, const float* x, float* dw, float* workspace, size_t workspace_bytes, int N, int Di,
                                int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co, int KD, int KH, int KW, int dstride,
                                int stride, int dpad, int pad, int accumulate, hipStream_t stream);

/* Kernels of the fp32 implicit-GEMM gather family (conv forward / data gradient / Linear) launched by this process so far.  One
 * C-ABI call may issue several: one per parity class of a strided data gradient, and main + fix-up launch when the tiles of a last
 * partial quantum (workgroup count just above a multiple of 256, or at most 128 in all) are K-split (csrc/conv_gemm.hip
 * gather_ksplit_plan, switch EDRL_GATHER_TAIL_SPLIT); bench.py reads the difference around a call so that its launch count equals
 * rocprofv3's.
 * The library allocates nothing: the slab the split workgroups hand their accumulators over in is the caller's, see below. */
long edrl_gather_launch_count(void);
/* How this build of the library multiplies fp32 operands in the conv / Linear contractions (forward, data and weight gradient):
 * 1 = every fp32 operand element is split exactly into three bf16 values and the product formed from six exact bf16 products on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation (libedrl_hip.so; csrc/conv_gemm.hip EDRL_F32_SPLIT: error against fp64 at or
 * below the fp32 MFMA's, tests/test_gpu_kernels.py); 0 = v_mfma_f32_32x32x2_f32 (libedrl_hip_f32mfma.so, same sources).
 * Replaces nothing in the reference (torch's conv2d / linear have one fp32 path); bench.py reports it. */
int edrl_f32_contraction_split(void);

/* K-split workspace of the fp32 gather family (csrc/conv_gemm.hip gather_ksplit_plan).  The caller registers ONE slab of at least
 * edrl_gather_ksplit_workspace_bytes() (16 MiB, 16-byte aligned) per (current HIP device, stream) it launches the family on; the
 * library keeps only the (device, stream) -> pointer table (64 entries: -28 when full) and never allocates or frees.  slab NULL
 * forgets the entry.  A launch on a (device, stream) without a registered slab runs unsplit: same body tiles bit for bit, tail
 * tiles differ by the association of the K sum -- register before the first launch if run-to-run identity across streams matters
 * (<package>/_lib.py does, from the torch caching allocator, the first time a stream issues a library call).  The slab must stay
 * valid until the stream's last gather launch has completed or the entry is replaced. */
size_t edrl_gather_ksplit_workspace_bytes(void);
int edrl_gather_ksplit_set_workspace(float* slab, size_t bytes, hipStream_t stream);

/* Run-time switches (EDRL_* environment variables, csrc/edrl_config.h) are read ONCE at first use; this re-reads them.  For tests
 * and A/B scripts, between launches (not while other threads launch).  Returns 1 if the library was built with -DEDRL_DIAG
 * (diagnostic kernel variants present: libedrl_hip_diag.so), 0 for the shipped library. */
int edrl_config_reload(void);

/* Measurement aid: an empty one-wave dispatch (kernel `edrl_trace_mark_kernel`).  The host-side call tracer
 * (<package>/_lib.py trace_begin / trace_end, scripts/step_trace.py) issues one in front of every library call so that a rocprofv3
 * kernel trace or --pmc pass of a whole training step can be attributed call by call (layer by layer) from dispatch order alone. */
int edrl_trace_mark(hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* EDRL_HIP_H */
