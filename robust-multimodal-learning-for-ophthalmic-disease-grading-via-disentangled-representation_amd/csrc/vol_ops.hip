// Volume (3-D) operators for the 3-D-conv OCT encoder, SURVEY.md §8(f) row 4 ("true 3D-conv OCT encoder").
//
// A k_d x k x k convolution with depth stride s_d over an NDHWC volume is computed as the 2-D implicit-GEMM convolution of
// conv_gemm.hip over the DEPTH-UNFOLDED volume: X'[n, do, h, w, kd*C + c] = X[n, do*s_d - p_d + kd, h, w, c] (zero outside),
// images = the N*Do output slices, channels = k_d*C.  The weight is stored [Co, KH, KW, KD*C (padded to a multiple of 4)], i.e.
// already in the K order of the 2-D kernel.  The unfold costs one extra streaming pass per conv (HBM-bound, k_d x the volume
// written); every MFMA contraction (forward, data gradient, weight gradient) then runs on the tuned 2-D kernels unchanged.
#include "edrl_common.h"
#include <stdint.h>

static inline int vgrid(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// y [N,Do,P,CK] <- x [N,D,P,C]   (P = H*W pixels per slice; channels j >= KD*C are zero padding)
__global__ __launch_bounds__(256) void depth_unfold_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int D,
                                                           int Do, long P, int C, int KD, int sd, int pd, int CK) {
  const long total = (long)N * Do * P * CK;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % CK);
    long t = i / CK;
    const long p = t % P;
    t /= P;
    const int dout = (int)(t % Do);
    const int n = (int)(t / Do);
    float v = 0.f;
    if (j < KD * C) {
      const int kd = j / C, c = j - kd * C;
      const int di = dout * sd - pd + kd;
      if (di >= 0 && di < D) v = x[(((long)n * D + di) * P + p) * C + c];
    }
    y[i] = v;
  }
}
// C % 4 == 0: one float4 per lane
__global__ __launch_bounds__(256) void depth_unfold_v4_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int D,
                                                              int Do, long P, int C, int KD, int sd, int pd) {
  const int C4 = C >> 2, CK4 = KD * C4;
  const long total = (long)N * Do * P * CK4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % CK4);
    long t = i / CK4;
    const long p = t % P;
    t /= P;
    const int dout = (int)(t % Do);
    const int n = (int)(t / Do);
    const int kd = j / C4, c4 = j - kd * C4;
    const int di = dout * sd - pd + kd;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (di >= 0 && di < D) v = *reinterpret_cast<const f32x4*>(x + (((long)n * D + di) * P + p) * C + c4 * 4);
    *reinterpret_cast<f32x4*>(y + i * 4) = v;
  }
}
// adjoint (gather form, no atomics): dx[n,di,p,c] = sum_{kd, do : do*sd - pd + kd == di} dy[n,do,p,kd*C + c]
__global__ __launch_bounds__(256) void depth_fold_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int D, int Do,
                                                         long P, int C, int KD, int sd, int pd, int CK) {
  const long total = (long)N * D * P * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const long p = t % P;
    t /= P;
    const int di = (int)(t % D);
    const int n = (int)(t / D);
    float s = 0.f;
    for (int kd = 0; kd < KD; ++kd) {
      const int num = di + pd - kd;
      if (num < 0 || num % sd) continue;
      const int dout = num / sd;
      if (dout < Do) s += dy[(((long)n * Do + dout) * P + p) * CK + kd * C + c];
    }
    dx[i] = s;
  }
}

// bf16 storage (the bf16 3-D trunk): 8 channels (16 bytes) per lane; the fold sums its <= KD taps in fp32 and rounds once
typedef unsigned short vol_u16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void depth_unfold_h8_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y,
                                                              int N, int D, int Do, long P, int C, int KD, int sd, int pd) {
  const int C8 = C >> 3, CK8 = KD * C8;
  const long total = (long)N * Do * P * CK8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int j = (int)(i % CK8);
    long t = i / CK8;
    const long p = t % P;
    t /= P;
    const int dout = (int)(t % Do);
    const int n = (int)(t / Do);
    const int kd = j / C8, c8 = j - kd * C8;
    const int di = dout * sd - pd + kd;
    vol_u16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (di >= 0 && di < D) v = *reinterpret_cast<const vol_u16x8*>(x + (((long)n * D + di) * P + p) * C + c8 * 8);
    *reinterpret_cast<vol_u16x8*>(y + i * 8) = v;
  }
}
__global__ __launch_bounds__(256) void depth_fold_h8_kernel(const unsigned short* __restrict__ dy, unsigned short* __restrict__ dx,
                                                            int N, int D, int Do, long P, int C, int KD, int sd, int pd) {
  const int C8 = C >> 3;
  const long CK = (long)KD * C;
  const long total = (long)N * D * P * C8;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long t = i / C8;
    const long p = t % P;
    t /= P;
    const int di = (int)(t % D);
    const int n = (int)(t / D);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int kd = 0; kd < KD; ++kd) {
      const int num = di + pd - kd;
      if (num < 0 || num % sd) continue;
      const int dout = num / sd;
      if (dout >= Do) continue;
      const vol_u16x8 v = *reinterpret_cast<const vol_u16x8*>(dy + (((long)n * Do + dout) * P + p) * CK + (long)kd * C + c8 * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += __uint_as_float((unsigned)v[e] << 16);
    }
    vol_u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {                    // round to nearest even (finite values; NaN stays NaN)
      const unsigned u = __float_as_uint(s[e]);
      o[e] = (s[e] != s[e]) ? (unsigned short)0x7fc0 : (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    *reinterpret_cast<vol_u16x8*>(dx + i * 8) = o;
  }
}

// max over depth, kernel 3 / stride 2 / pad 1 (the depth half of MaxPool3d(3, 2, 1); the spatial half is the 2-D max-pool of
// bn_pool.hip applied per slice -- max is separable).  idx: which of the 3 taps won (first maximum, like torch).
__global__ __launch_bounds__(256) void maxpool_depth_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                unsigned char* __restrict__ idx, int N, int D, int Do, long PC) {
  const long total = (long)N * Do * PC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long q = i % PC;
    const long t = i / PC;
    const int dout = (int)(t % Do);
    const int n = (int)(t / Do);
    float best = -INFINITY;
    int bi = 0;
    for (int k = 0; k < 3; ++k) {
      const int di = 2 * dout - 1 + k;
      if (di < 0 || di >= D) continue;
      const float v = x[((long)n * D + di) * PC + q];
      if (v > best || (v != v && best == best)) { best = v; bi = k; }
    }
    y[i] = best;
    idx[i] = (unsigned char)bi;
  }
}
__global__ __launch_bounds__(256) void maxpool_depth_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                                float* __restrict__ dx, int N, int D, int Do, long PC) {
  const long total = (long)N * D * PC;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long q = i % PC;
    const long t = i / PC;
    const int di = (int)(t % D);
    const int n = (int)(t / D);
    float s = 0.f;
    for (int k = 0; k < 3; ++k) {           // outputs do with 2*do - 1 + k == di
      const int num = di + 1 - k;
      if (num < 0 || (num & 1)) continue;
      const int dout = num >> 1;
      if (dout >= Do) continue;
      const long o = ((long)n * Do + dout) * PC + q;
      if (idx[o] == k) s += dy[o];
    }
    dx[i] = s;
  }
}

extern "C" {

int edrl_depth_unfold_f32(const float* x, float* y, int N, int D, long P, int C, int KD, int sd, int pd, int Do, int CK,
                          hipStream_t st) {
  if (N <= 0 || D <= 0 || P <= 0 || C <= 0 || KD <= 0 || sd <= 0 || pd < 0 || Do <= 0 || CK < KD * C) return EDRL_EINVAL;
  if ((C & 3) == 0 && CK == KD * C && ((((uintptr_t)x | (uintptr_t)y) & 15) == 0))
    hipLaunchKernelGGL(depth_unfold_v4_kernel, dim3(vgrid((long)N * Do * P * (CK / 4))), dim3(256), 0, st, x, y, N, D, Do, P, C,
                       KD, sd, pd);
  else
    hipLaunchKernelGGL(depth_unfold_kernel, dim3(vgrid((long)N * Do * P * CK)), dim3(256), 0, st, x, y, N, D, Do, P, C, KD, sd,
                       pd, CK);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_depth_fold_f32(const float* dy, float* dx, int N, int D, long P, int C, int KD, int sd, int pd, int Do, int CK,
                        hipStream_t st) {
  if (N <= 0 || D <= 0 || P <= 0 || C <= 0 || KD <= 0 || sd <= 0 || pd < 0 || Do <= 0 || CK < KD * C) return EDRL_EINVAL;
  hipLaunchKernelGGL(depth_fold_kernel, dim3(vgrid((long)N * D * P * C)), dim3(256), 0, st, dy, dx, N, D, Do, P, C, KD, sd, pd,
                     CK);
  EDRL_LAUNCH_CHECK();
  return 0;
}
// bf16 tensors (C % 8 == 0, CK == KD*C, 16-byte aligned): the depth-unfolded operand / the fold of its gradient for the bf16 3-D trunk
int edrl_depth_unfold_bf16(const void* x, void* y, int N, int D, long P, int C, int KD, int sd, int pd, int Do, hipStream_t st) {
  if (N <= 0 || D <= 0 || P <= 0 || C <= 0 || (C & 7) || KD <= 0 || sd <= 0 || pd < 0 || Do <= 0 || !x || !y) return EDRL_EINVAL;
  if ((((uintptr_t)x | (uintptr_t)y) & 15) != 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(depth_unfold_h8_kernel, dim3(vgrid((long)N * Do * P * (KD * (C / 8)))), dim3(256), 0, st,
                     (const unsigned short*)x, (unsigned short*)y, N, D, Do, P, C, KD, sd, pd);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_depth_fold_bf16(const void* dy, void* dx, int N, int D, long P, int C, int KD, int sd, int pd, int Do, hipStream_t st) {
  if (N <= 0 || D <= 0 || P <= 0 || C <= 0 || (C & 7) || KD <= 0 || sd <= 0 || pd < 0 || Do <= 0 || !dy || !dx) return EDRL_EINVAL;
  if ((((uintptr_t)dy | (uintptr_t)dx) & 15) != 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(depth_fold_h8_kernel, dim3(vgrid((long)N * D * P * (C / 8))), dim3(256), 0, st, (const unsigned short*)dy,
                     (unsigned short*)dx, N, D, Do, P, C, KD, sd, pd);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_maxpool_depth3s2_fwd_f32(const float* x, float* y, unsigned char* idx, int N, int D, long PC, hipStream_t st) {
  if (N <= 0 || D <= 0 || PC <= 0) return EDRL_EINVAL;
  const int Do = (D + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_depth_fwd_kernel, dim3(vgrid((long)N * Do * PC)), dim3(256), 0, st, x, y, idx, N, D, Do, PC);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_maxpool_depth3s2_bwd_f32(const float* dy, const unsigned char* idx, float* dx, int N, int D, long PC, hipStream_t st) {
  if (N <= 0 || D <= 0 || PC <= 0) return EDRL_EINVAL;
  const int Do = (D + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool_depth_bwd_kernel, dim3(vgrid((long)N * D * PC)), dim3(256), 0, st, dy, idx, dx, N, D, Do, PC);
  EDRL_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
