// Shared device helpers for the EDRL gfx950 (CDNA4) kernels.
// gfx950 only: 64-wide wavefronts, MFMA matrix cores, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define EDRL_WAVE 64

// Error convention of the C-ABI: 0 = ok, otherwise a hipError_t value, or a
// negative EDRL_E* code for argument violations detected on the host.
#define EDRL_EINVAL (-22)
#define EDRL_ENOSPC (-28)

#define EDRL_LAUNCH_CHECK()                      \
  do {                                           \
    hipError_t e__ = hipGetLastError();          \
    if (e__ != hipSuccess) return (int)e__;      \
  } while (0)

// BatchNorm forms shared by the kernels that apply them on the fly.  Next to the fp32 MFMA every VALU instruction costs
// matrix-pipe time (both run on the SIMD's fp32 lanes), so the fused operand transforms are single packed FMAs:
//   activation   relu(x*scale + shift2),  shift2 = shift - mean*scale          (fcoef rows 2 and 4)
//   d_raw        A*g + nK2*x + C2,        nK2 = -A*rstd*mean(g*xhat), C2 = -nK2*mean - A*mean(g)   (bcoef rows 0..2)
// The ReLU decision recomputed in backward uses the identical expression, hence the identical bits.
__device__ __forceinline__ float edrl_bn_pre(float x, float mean, float scale, float shift) {
  return __builtin_fmaf(x - mean, scale, shift);
}
__device__ __forceinline__ f32x4 edrl_bn_pre2(f32x4 x, f32x4 scale, f32x4 shift2) {
  return __builtin_elementwise_fma(x, scale, shift2);
}
// NaN semantics: v_max_f32 is IEEE maxNum, so relu(NaN) = 0 here whereas torch.relu propagates the NaN.  A NaN / Inf in a raw conv
// output makes that channel's batch statistics -- hence scale / shift2 and the running statistics -- non-finite, and the fused
// paths then feed zeros downstream instead of NaNs.  The divergence is therefore caught where it is still visible: the running
// statistics (train() / bench.py check them, MedFusion.raise_on_nonfinite), not in the K loop (a propagating form is two more
// VALU instructions per element next to the fp32 MFMA).
__device__ __forceinline__ f32x4 edrl_bn_relu2(f32x4 x, f32x4 scale, f32x4 shift2) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  return __builtin_elementwise_max(edrl_bn_pre2(x, scale, shift2), z);
}
__device__ __forceinline__ f32x4 edrl_bn_bwd_dx2(f32x4 g, f32x4 x, f32x4 A, f32x4 nK2, f32x4 C2) {
  return __builtin_elementwise_fma(nK2, x, __builtin_elementwise_fma(A, g, C2));
}

static inline int edrl_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// XCD-aware bijective block remap (8 XCDs, blocks are dealt round-robin):
// blocks that share an XCD get a contiguous range of logical tile ids so
// that tiles sharing operand panels hit the same 4 MiB L2.
__device__ __forceinline__ int edrl_xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + slot;
}

__device__ __forceinline__ float edrl_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double edrl_wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float edrl_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x == 256 (4 waves). `sh` must hold >= 4 floats.
__device__ __forceinline__ float edrl_block_sum_256(float v, float* sh) {
  v = edrl_wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ double edrl_block_sum_256_d(double v, double* sh) {
  v = edrl_wave_sum_d(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  return sh[0] + sh[1] + sh[2] + sh[3];
}
