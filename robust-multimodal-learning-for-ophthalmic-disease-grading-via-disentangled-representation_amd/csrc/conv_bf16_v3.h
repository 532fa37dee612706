// Launch interface of the 256x256 LDS-DMA bf16 implicit-GEMM core (conv_bf16_v3.hip), used by the C-ABI launchers in conv_bf16.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_geom.h"

// true when the geometry can run on the v3 core (channel counts, descriptor footprints) AND is expected to be faster there.
bool gather_bf16_v3_ok(const GatherGeom& g, bool dgrad);
// forward (dgrad = false) or data gradient (one parity class per call) of the plain bf16 convolution; honours GF_STATS / GF_ACCUM.
// `fuse` (optional, data gradient only): fuse->ep_x != NULL selects the BatchNorm-backward epilogue (result masked with the ReLU
// decision of the BatchNorm below, partial sums to fuse->ep_part; see the kernel); gather_bf16_v3_epi_ok checks its operands.
bool gather_bf16_v3_epi_ok(const GatherGeom& g, const GatherFuse& F);
int launch_gather_bf16_v3(const void* src, const void* wm, void* dst, const GatherGeom& g, bool dgrad, hipStream_t st,
                          const GatherFuse* fuse = nullptr);

// Plain bf16 weight gradient on the 256x256 LDS-DMA core (conv_wgrad_bf16_v3.hip): writes `*splits_out` fp32 partial slabs
// [split][Co][KH*KW*Ci] into `workspace`; the caller reduces them (splitk_reduce_h_kernel).
bool wgrad_bf16_v3_ok(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad);
size_t wgrad_bf16_v3_workspace_bytes(int N, int Ho, int Wo, int Co, int Ci, int KH, int KW);
int launch_wgrad_bf16_v3(const void* dy, const void* x, float* workspace, size_t workspace_bytes, int N, int Hi, int Wi, int Ci,
                         int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, int* splits_out, hipStream_t st);

// 3x3 / stride 1 / pad 1 between two 64-channel tensors on the weight-stationary kernel of conv_c64_bf16.hip (forward, flip = 0, with
// optional BatchNorm chunk partials; data gradient, flip = 1 on the permuted weights, optionally with the masked-gradient epilogue).
bool conv3x3_c64_ok(int N, int H, int W, int Ci, int Co, int KH, int KW, int stride, int pad);
int launch_conv3x3_c64(const void* src, const void* w, int flip, void* dst, int N, int H, int W, float* stat_part, const void* ep_x,
                       const unsigned char* ep_mask, float* ep_part, const float* ep_mean, hipStream_t st);

// 1x1 / stride 1 convolution with 64 or 128 input channels and 64 .. 512 output channels on the streaming kernel of conv_c64_bf16.hip
// (in_fcoef != NULL: BatchNorm + ReLU of the input folded into the operand, fcoef [5][64]); stat_part optional.
bool conv1x1_k64_ok(int N, int H, int W, int Ci, int Co, int KH, int KW, int stride, int pad);
int launch_conv1x1_k64(const void* x, const float* in_fcoef, const void* w, void* y, int N, int H, int W, int Ci, int Co,
                       float* stat_part, hipStream_t st);

// Stem of the 1-channel bf16 trunk on the streaming kernel of conv_c64_bf16.hip (ATR 2): 4x4 / pad 2 convolution over the fp32
// space-to-depth image, operands rounded to bf16 in registers, bf16 MFMA, fp32 accumulate, bf16 output + BatchNorm chunk partials.
bool stem_s2d_bf16_ok(int N, int Hs, int Ws);
int launch_stem_s2d_bf16(const float* xs, const void* w, void* y, int N, int Hs, int Ws, float* stat_part, hipStream_t st);

// Backward of the expanding 1x1 layers of the first residual stage (64 -> 256 channels) as one streaming kernel
// (conv1x1_bwd_bf16.hip): weight gradient slabs [splits][256][64] + masked data gradient + BatchNorm-backward partial sums.
bool conv1x1_k64_bwd_ok(int N, int H, int W, int Ci, int Co);
long conv1x1_k64_bwd_chunks(int N, int H, int W);
size_t conv1x1_k64_bwd_workspace_bytes(int N, int H, int W);
int conv1x1_k64_bwd_splits(int N, int H, int W);
int launch_conv1x1_k64_bwd(const void* g, const void* yraw, const float* bcoef, const void* x2, const float* x2coef, const void* wt,
                           void* g2, float* ep_part, float* dw_slabs, int N, int H, int W, hipStream_t st);

// Persistent form of the core (conv_bf16_v3p.hip: next tile's first units issued before the epilogue, register epilogue): plain
// forward / data gradient (GF_STATS, GF_ACCUM, strided classes); same geometry conditions as gather_bf16_v3_ok.
bool gather_bf16_v3p_ok(const GatherGeom& g, bool epi);
int launch_gather_bf16_v3p(const void* src, const void* wm, void* dst, const GatherGeom& g, bool dgrad, hipStream_t st,
                           const GatherFuse* fuse = nullptr);

// Weight gradient of the same stem (bf16 d_raw x fp32 space-to-depth image -> fp32 [64][4x4x4] slabs), conv_c64_bf16.hip.
int stem_wgrad_s2d_bf16_splits(int N, int Hs, int Ws);
size_t stem_wgrad_s2d_bf16_workspace_bytes(int N, int Hs, int Ws);
int launch_stem_wgrad_s2d_bf16(const void* dy, const float* xs, float* slabs, int N, int Hs, int Ws, hipStream_t st);

// Small-tile form of the core (conv_bf16_v3s.hip: 128 x 128 tile, 4 waves, 64 KiB ring, two workgroups per CU): plain forward / data
// gradient (GF_STATS, GF_ACCUM, strided classes) and the data gradient with the BatchNorm-backward epilogue (fuse->ep_x != NULL).
// gather_bf16_v3s_can: the geometry can run there (SC % 32, NC % 128, descriptor footprints); which layers SHOULD is decided in
// conv_bf16.hip (EDRL_BF16_V3S).
bool gather_bf16_v3s_can(const GatherGeom& g);
int launch_gather_bf16_v3s(const void* src, const void* wm, void* dst, const GatherGeom& g, bool dgrad, hipStream_t st,
                           const GatherFuse* fuse = nullptr);
