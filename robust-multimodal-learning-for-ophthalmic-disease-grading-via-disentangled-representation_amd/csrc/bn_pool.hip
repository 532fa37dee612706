// BatchNorm (train mode) statistics / apply / backward, pooling and layout kernels, NHWC fp32.
// HBM-bound wavefront-reduction kernels with 16 B/lane coalesced accesses.
//
// Covers SURVEY.md §8(a) rows E3 (BN2d / ReLU / residual / maxpool / avgpool inside the
// build-owned encoders behind fusion_net.py:884-885) and H13 (DILR.bn1/bn2,
// fusion_net.py:653-654,658,757-758: BatchNorm1d(2048, affine=False), momentum 0.1, eps 1e-5,
// biased variance to normalise, unbiased variance into running_var).
//
// Reductions are two-level and ordered (per-chunk fp32 partials, fp64 combine): deterministic,
// no atomics.
#include "edrl_common.h"
#include "edrl_config.h"

#define BN_ROWS_PER_CHUNK 1024

// Typed 4-channel accessors: the same kernels serve fp32 tensors and the bf16 activations / gradients of the C2 path
// (bf16 storage, fp32 statistics and arithmetic).
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<__bf16>(const __bf16* p) {
  const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
  f32x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (float)v[e];
  return o;
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* p, f32x4 v) {
  bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
  *reinterpret_cast<bf16x4*>(p) = o;
}

// MODE 0: shifted moments (sum (x-K), sum (x-K)^2, K) with K = the chunk's first row: one pass, and no
//         catastrophic cancellation when var << mean^2 (e.g. BatchNorm1d over 2 rows).
// MODE 1: (sum g, sum g*xhat) with g = dout * (out > 0 if out given).
// Partial layout: part[chunk][3][C].
template <int MODE, typename TR, typename TA>
__global__ __launch_bounds__(256) void colstat_kernel(const TR* __restrict__ x, const TA* __restrict__ dout,
                                                      const TA* __restrict__ out,
                                                      const unsigned char* __restrict__ rmask,
                                                      const float* __restrict__ mean,
                                                      const float* __restrict__ rstd, long M, int C, long ld,
                                                      float* __restrict__ part, TA* __restrict__ gout = nullptr) {
  __shared__ float sh[256 * 8];
  const int C4 = C >> 2;
  const int CG = C4 < 64 ? C4 : 64;  // float4 column groups per block (power of two)
  const int RL = 256 / CG;
  const int tid = threadIdx.x;
  const int cg = tid % CG, rl = tid / CG;
  const int c = (blockIdx.y * 64 + cg) * 4;
  const long row0 = (long)blockIdx.x * BN_ROWS_PER_CHUNK;
  long row1 = row0 + BN_ROWS_PER_CHUNK;
  if (row1 > M) row1 = M;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 1) {
      mu = *reinterpret_cast<const f32x4*>(mean + c);
      rs = *reinterpret_cast<const f32x4*>(rstd + c);
    } else {
      mu = ld4<TR>(x + row0 * ld + c);  // shift K
    }
    // 4 rows per iteration: 4 (MODE 0) or 12 (MODE 1) independent 16-B loads in flight per lane
    constexpr int U = 4;
    long r = row0 + rl;
    if (rl < RL) {
      for (; r + (long)(U - 1) * RL < row1; r += (long)U * RL) {
        f32x4 xv[U], gv[U], ov[U];
#pragma unroll
        for (int u = 0; u < U; ++u) xv[u] = ld4<TR>(x + (r + (long)u * RL) * ld + c);
        if (MODE == 1) {
#pragma unroll
          for (int u = 0; u < U; ++u) gv[u] = ld4<TA>(dout + (r + (long)u * RL) * ld + c);
          if (rmask) {
#pragma unroll
            for (int u = 0; u < U; ++u) ov[u][0] = (float)rmask[(r + (long)u * RL) * C4 + (c >> 2)];
          } else if (out) {
#pragma unroll
            for (int u = 0; u < U; ++u) ov[u] = ld4<TA>(out + (r + (long)u * RL) * ld + c);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (MODE == 0) {
            const f32x4 d = xv[u] - mu;
            s0 += d;
            s1 += d * d;
          } else {
            f32x4 g = gv[u];
            if (rmask) {
              const int mb = (int)ov[u][0];
#pragma unroll
              for (int e = 0; e < 4; ++e) g[e] = (mb >> e) & 1 ? g[e] : 0.f;
            } else if (out) {
#pragma unroll
              for (int e = 0; e < 4; ++e) g[e] = ov[u][e] > 0.f ? g[e] : 0.f;
            }
            s0 += g;
            s1 += g * ((xv[u] - mu) * rs);
            if (gout) st4<TA>(gout + (r + (long)u * RL) * ld + c, g);
          }
        }
      }
      for (; r < row1; r += RL) {
        const f32x4 xv = ld4<TR>(x + r * ld + c);
        if (MODE == 0) {
          const f32x4 d = xv - mu;
          s0 += d;
          s1 += d * d;
        } else {
          f32x4 g = ld4<TA>(dout + r * ld + c);
          if (rmask) {
            const int mb = rmask[r * C4 + (c >> 2)];
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = (mb >> e) & 1 ? g[e] : 0.f;
          } else if (out) {
            const f32x4 o = ld4<TA>(out + r * ld + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
          }
          s0 += g;
          s1 += g * ((xv - mu) * rs);
          if (gout) st4<TA>(gout + r * ld + c, g);
        }
      }
    }
  }
  float* my = sh + tid * 8;
#pragma unroll
  for (int e = 0; e < 4; ++e) { my[e] = s0[e]; my[4 + e] = s1[e]; }
  __syncthreads();
  if (rl == 0 && c < C) {
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    for (int q = 0; q < RL; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += sh[(q * CG + cg) * 8 + e];
    float* p = part + (long)blockIdx.x * 3 * C;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      p[c + e] = a[e]; p[C + c + e] = a[4 + e];
      if (MODE == 0) p[2 * C + c + e] = (float)x[row0 * ld + c + e];
    }
  }
}

// Combine chunk partials in fp64. block = 16 channels x 16 chunk lanes.
#define FIN_CH 16
#define FIN_LANES 16
__device__ __forceinline__ double fin_lane_sum(double v, double* sh) {
  const int t = threadIdx.x, cc = t & (FIN_CH - 1);
  __syncthreads();
  sh[t] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int q = 0; q < FIN_LANES; ++q) s += sh[q * FIN_CH + cc];
  return s;
}
__device__ __forceinline__ void combine_partials(const float* part, int nchunks, int C, int c, int q,
                                                 double* sh, double& s0, double& s1) {
  double a0 = 0.0, a1 = 0.0;
  if (c < C)
    for (int k = q; k < nchunks; k += FIN_LANES) {
      a0 += (double)part[(long)k * 3 * C + c];
      a1 += (double)part[(long)k * 3 * C + C + c];
    }
  s0 = fin_lane_sum(a0, sh);
  s1 = fin_lane_sum(a1, sh);
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ part, int nchunks, int C,
                                                          long M, int rows_per_chunk,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta,
                                                          float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float momentum,
                                                          float eps, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out, float* __restrict__ scale,
                                                          float* __restrict__ shift, float* __restrict__ shift2 = nullptr) {
  __shared__ double sh[256];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), q = threadIdx.x / FIN_CH;
  // pass 1: mean = sum_c (n_c K_c + S1_c) / M
  double a0 = 0.0;
  if (c < C)
    for (int k = q; k < nchunks; k += FIN_LANES) {
      const long r0 = (long)k * rows_per_chunk;
      const double n = (double)((M - r0) < rows_per_chunk ? (M - r0) : rows_per_chunk);
      a0 += n * (double)part[(long)k * 3 * C + 2 * C + c] + (double)part[(long)k * 3 * C + c];
    }
  const double mu = fin_lane_sum(a0, sh) / (double)M;
  // pass 2: M2 = sum_c [ S2_c - 2 (mu-K_c) S1_c + n_c (mu-K_c)^2 ]
  double a1 = 0.0;
  if (c < C)
    for (int k = q; k < nchunks; k += FIN_LANES) {
      const long r0 = (long)k * rows_per_chunk;
      const double n = (double)((M - r0) < rows_per_chunk ? (M - r0) : rows_per_chunk);
      const double dk = mu - (double)part[(long)k * 3 * C + 2 * C + c];
      a1 += (double)part[(long)k * 3 * C + C + c] - 2.0 * dk * (double)part[(long)k * 3 * C + c] + n * dk * dk;
    }
  const double m2 = fin_lane_sum(a1, sh);
  if (q == 0 && c < C) {
    double var = m2 / (double)M;
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    const float muf = (float)mu;
    mean_out[c] = muf;
    rstd_out[c] = rs;
    const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
    const float sc = gm * rs;
    scale[c] = sc;
    shift[c] = bt;  // applied as (x - mean)*scale + shift: subtract first, no cancellation
    if (shift2) shift2[c] = (float)((double)bt - mu * (double)sc);   // single-FMA form x*scale + shift2 of the fused conv loads
    if (running_mean) {
      const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * muf;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
    }
  }
}

// Two-stage variant for large chunk counts (the conv-epilogue partials have one chunk per 128 rows: up to ~10^5):
// stage A reduces groups of FIN_GROUP chunks to (n_g, mean_g, M2_g) in fp64 with many workgroups, stage B merges the
// groups (Chan et al.) and finalises.  Same arithmetic as bn_finalize_kernel, just more parallel.
#define FIN_GROUP 64
__global__ __launch_bounds__(256) void bn_group_kernel(const float* __restrict__ part, int nchunks, int C, long M,
                                                       int rows_per_chunk, double* __restrict__ gout) {
  __shared__ double sh[256];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), q = threadIdx.x / FIN_CH;
  const int k0 = blockIdx.y * FIN_GROUP;
  int k1 = k0 + FIN_GROUP;
  if (k1 > nchunks) k1 = nchunks;
  long rbeg = (long)k0 * rows_per_chunk, rend = (long)k1 * rows_per_chunk;
  if (rend > M) rend = M;
  const double ng = (double)(rend - rbeg);
  double a0 = 0.0;
  if (c < C)
    for (int k = k0 + q; k < k1; k += FIN_LANES) {
      const long r0 = (long)k * rows_per_chunk;
      const double n = (double)((M - r0) < rows_per_chunk ? (M - r0) : rows_per_chunk);
      a0 += n * (double)part[(long)k * 3 * C + 2 * C + c] + (double)part[(long)k * 3 * C + c];
    }
  const double mu = fin_lane_sum(a0, sh) / ng;
  double a1 = 0.0;
  if (c < C)
    for (int k = k0 + q; k < k1; k += FIN_LANES) {
      const long r0 = (long)k * rows_per_chunk;
      const double n = (double)((M - r0) < rows_per_chunk ? (M - r0) : rows_per_chunk);
      const double dk = mu - (double)part[(long)k * 3 * C + 2 * C + c];
      a1 += (double)part[(long)k * 3 * C + C + c] - 2.0 * dk * (double)part[(long)k * 3 * C + c] + n * dk * dk;
    }
  const double m2 = fin_lane_sum(a1, sh);
  if (q == 0 && c < C) {
    double* o = gout + (long)blockIdx.y * 3 * C;
    o[c] = ng; o[C + c] = mu; o[2 * C + c] = m2;
  }
}
__global__ __launch_bounds__(256) void bn_finalize_groups_kernel(const double* __restrict__ gin, int G, int C, long M,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta,
                                                                float* __restrict__ running_mean,
                                                                float* __restrict__ running_var, float momentum,
                                                                float eps, float* __restrict__ mean_out,
                                                                float* __restrict__ rstd_out, float* __restrict__ scale,
                                                                float* __restrict__ shift, float* __restrict__ shift2 = nullptr) {
  // 16 channels x 16 lanes per workgroup: a lane walks every 16th group (several hundred groups for the 56x56 layers:
  // one thread per channel was a ~50 us chain of dependent loads on the critical path of every conv+BN pair)
  __shared__ double sh[256];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), q = threadIdx.x / FIN_CH;
  double a0 = 0.0;
  if (c < C)
    for (int gI = q; gI < G; gI += FIN_LANES) a0 += gin[(long)gI * 3 * C + c] * gin[(long)gI * 3 * C + C + c];
  const double mu = fin_lane_sum(a0, sh) / (double)M;
  double a1 = 0.0;
  if (c < C)
    for (int gI = q; gI < G; gI += FIN_LANES) {
      const double d = gin[(long)gI * 3 * C + C + c] - mu;
      a1 += gin[(long)gI * 3 * C + 2 * C + c] + gin[(long)gI * 3 * C + c] * d * d;
    }
  const double m2 = fin_lane_sum(a1, sh);
  if (q != 0 || c >= C) return;
  double var = m2 / (double)M;
  if (var < 0.0) var = 0.0;
  const float rs = (float)(1.0 / sqrt(var + (double)eps));
  const float muf = (float)mu;
  mean_out[c] = muf;
  rstd_out[c] = rs;
  const float gm = gamma ? gamma[c] : 1.f, bt = beta ? beta[c] : 0.f;
  scale[c] = gm * rs;
  shift[c] = bt;
  if (shift2) shift2[c] = (float)((double)bt - mu * (double)(gm * rs));
  if (running_mean) {
    const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * muf;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unb;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int nchunks, int C,
                                                              long M, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate,
                                                              float* __restrict__ coef) {
  __shared__ double sh[256];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), q = threadIdx.x / FIN_CH;
  double s0, s1;
  combine_partials(part, nchunks, C, c, q, sh, s0, s1);
  if (q == 0 && c < C) {
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s0 : (float)s0;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)s1 : (float)s1;
    coef[c] = (float)(s0 / (double)M);
    coef[C + c] = (float)(s1 / (double)M);
  }
}

// ---- BatchNorm backward finalize from per-tile partial sums part[chunk][planes][C] (plane 0 = sum g, plane 1 = sum g*xhat;
// the fused data-gradient epilogue emits planes = 2 with one chunk per 128 rows, colstat_kernel<1> planes = 3 per 1024 rows).
// With planes = 2 plane 1 is sum g*(x - mean) (the epilogues subtract the batch mean before the product, so no mean*sum(g) is
// cancelled afterwards; sum g*xhat = rstd * plane 1, in fp64).
// -> dgamma, dbeta and bcoef [4][C] = {A = gamma*rstd, nK2 = -A*rstd*mean(g*xhat), C2 = -nK2*mean - A*mean(g), mean}: the
// coefficients from which the consumers (conv_gemm.hip ATR 2 / DYT 2) form d_raw = A*g + nK2*x + C2 with two packed FMAs.
__global__ __launch_bounds__(256) void bn_bwd_group_kernel(const float* __restrict__ part, int nchunks, int planes, int C,
                                                           double* __restrict__ gout) {
  __shared__ double sh[256];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), q = threadIdx.x / FIN_CH;
  const int k0 = blockIdx.y * FIN_GROUP;
  int k1 = k0 + FIN_GROUP;
  if (k1 > nchunks) k1 = nchunks;
  double a0 = 0.0, a1 = 0.0;
  if (c < C)
    for (int k = k0 + q; k < k1; k += FIN_LANES) {
      a0 += (double)part[((long)k * planes) * C + c];
      a1 += (double)part[((long)k * planes + 1) * C + c];
    }
  const double s0 = fin_lane_sum(a0, sh), s1 = fin_lane_sum(a1, sh);
  if (q == 0 && c < C) {
    gout[((long)blockIdx.y * 2) * C + c] = s0;
    gout[((long)blockIdx.y * 2 + 1) * C + c] = s1;
  }
}
// groups == nullptr: reduce the float partials directly (few chunks)
__global__ __launch_bounds__(256) void bn_bwd_finalize_coef_kernel(const float* __restrict__ part, int nchunks, int planes,
                                                                   const double* __restrict__ groups, int G, int C, long M,
                                                                   const float* __restrict__ gamma,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                   float* __restrict__ bcoef) {
  __shared__ double sh[256];
  const int c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), q = threadIdx.x / FIN_CH;
  double a0 = 0.0, a1 = 0.0;
  if (c < C) {
    if (groups) {
      for (int k = q; k < G; k += FIN_LANES) { a0 += groups[((long)k * 2) * C + c]; a1 += groups[((long)k * 2 + 1) * C + c]; }
    } else {
      for (int k = q; k < nchunks; k += FIN_LANES) {
        a0 += (double)part[((long)k * planes) * C + c];
        a1 += (double)part[((long)k * planes + 1) * C + c];
      }
    }
  }
  const double s0 = fin_lane_sum(a0, sh);
  double s1 = fin_lane_sum(a1, sh);
  if (q != 0 || c >= C) return;
  const double rs = (double)rstd[c], mu = (double)mean[c];
  if (planes == 2) s1 = rs * s1;                  // the conv epilogue accumulates sum g*(x - mean): -> sum g*xhat
  if (dbeta) dbeta[c] = (float)s0;
  if (dgamma) dgamma[c] = (float)s1;
  const double A = (double)(gamma ? gamma[c] : 1.f) * rs;
  const double nK2 = -A * rs * (s1 / (double)M);
  bcoef[c] = (float)A;
  bcoef[C + c] = (float)nK2;
  bcoef[2 * (long)C + c] = (float)(-nK2 * mu - A * (s0 / (double)M));
  bcoef[3 * (long)C + c] = mean[c];
}

// out = act((x-mean)*scale + shift + residual)
// res_coef (optional) [5][C] = {mean, rstd, scale, shift, shift2}: the residual operand is itself a RAW conv output (the downsample
// branch) and its BatchNorm is applied here, so the branch's normalised copy is never stored.
template <typename TR, typename TA>
__global__ __launch_bounds__(256) void bn_apply_kernel(const TR* __restrict__ x, const float* __restrict__ mean,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift,
                                                       const TA* __restrict__ residual,
                                                       TA* __restrict__ out, unsigned char* __restrict__ mask_out,
                                                       long M, int C, long ld, int relu,
                                                       const float* __restrict__ res_coef = nullptr) {
  const int C4 = C >> 2;
  const long total = M * C4;
  const long gstride = (long)gridDim.x * blockDim.x;
  // Fast path (every ResNet layer): C/4 a power of two that divides the grid stride -> a thread keeps ONE column group for the
  // whole loop: coefficients loaded once, row index by shift (no 64-bit division per element), two elements in flight per trip.
  if ((C4 & (C4 - 1)) == 0 && (gstride & (C4 - 1)) == 0) {
    const int lg = __builtin_ctz(C4);
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = (int)(i & (C4 - 1)) * 4;
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
    const f32x4 sf = *reinterpret_cast<const f32x4*>(shift + c);
    f32x4 rm = {0.f, 0.f, 0.f, 0.f}, rsc = {1.f, 1.f, 1.f, 1.f}, rsf = {0.f, 0.f, 0.f, 0.f};
    if (residual && res_coef) {
      rm = *reinterpret_cast<const f32x4*>(res_coef + c);
      rsc = *reinterpret_cast<const f32x4*>(res_coef + 2 * (long)C + c);
      rsf = *reinterpret_cast<const f32x4*>(res_coef + 3 * (long)C + c);
    }
    auto one = [&](long ii, const f32x4 xv, f32x4 rv) {
      f32x4 v;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = edrl_bn_pre(xv[e], mu[e], sc[e], sf[e]);
      if (residual) {
        if (res_coef) {
#pragma unroll
          for (int e = 0; e < 4; ++e) rv[e] = edrl_bn_pre(rv[e], rm[e], rsc[e], rsf[e]);
        }
        v += rv;
      }
      if (relu) {
        if (mask_out) {
          int mb = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) mb |= (v[e] > 0.f ? 1 : 0) << e;
          if (C4 >= 4) {      // 4 neighbouring lanes hold 4 consecutive mask bytes (i, total and the grid stride are multiples of 4):
            const int b1 = __shfl_down(mb, 1, 64), b2 = __shfl_down(mb, 2, 64), b3 = __shfl_down(mb, 3, 64);   // one dword store
            if ((threadIdx.x & 3) == 0) *reinterpret_cast<unsigned int*>(mask_out + ii) = (unsigned)(mb | (b1 << 8) | (b2 << 16) | (b3 << 24));
          } else {
            mask_out[ii] = (unsigned char)mb;
          }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      st4<TA>(out + (ii >> lg) * ld + c, v);
    };
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    if (C4 <= 256) {
      // Block-contiguous chunks of 4 x 256 float4 (16 KiB per operand, all 8 loads of a thread in flight before the first use) --
      // the shape torch's own vectorised elementwise kernels stream at ~6 TB/s for a 2-read : 1-write mix on this GPU
      // (scripts/hbm_probe2.py); the column group (i & (C4-1)) is the thread's for every chunk because C4 divides 256.
      const long cstride = (long)gridDim.x * 1024;
      for (long base = (long)blockIdx.x * 1024; base < total; base += cstride) {
        long ii[4]; f32x4 xv[4], rv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          ii[u] = base + u * 256 + threadIdx.x;
          const bool ok = ii[u] < total;
          xv[u] = ok ? ld4<TR>(x + (ii[u] >> lg) * ld + c) : z4;
          rv[u] = (ok && residual) ? ld4<TA>(residual + (ii[u] >> lg) * ld + c) : z4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (ii[u] < total) one(ii[u], xv[u], rv[u]);
      }
      return;
    }
    for (; i + gstride < total; i += 2 * gstride) {
      const long j = i + gstride;
      const f32x4 x0 = ld4<TR>(x + (i >> lg) * ld + c), x1 = ld4<TR>(x + (j >> lg) * ld + c);
      const f32x4 r0 = residual ? ld4<TA>(residual + (i >> lg) * ld + c) : z4;
      const f32x4 r1 = residual ? ld4<TA>(residual + (j >> lg) * ld + c) : z4;
      one(i, x0, r0);
      one(j, x1, r1);
    }
    if (i < total) one(i, ld4<TR>(x + (i >> lg) * ld + c), residual ? ld4<TA>(residual + (i >> lg) * ld + c) : z4);
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gstride) {
    const long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    const f32x4 xv = ld4<TR>(x + r * ld + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + c);
    const f32x4 sf = *reinterpret_cast<const f32x4*>(shift + c);
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = edrl_bn_pre(xv[e], mu[e], sc[e], sf[e]);
    if (residual) {
      f32x4 rv = ld4<TA>(residual + r * ld + c);
      if (res_coef) {
        const f32x4 rm = *reinterpret_cast<const f32x4*>(res_coef + c);
        const f32x4 rsc = *reinterpret_cast<const f32x4*>(res_coef + 2 * (long)C + c);
        const f32x4 rsf = *reinterpret_cast<const f32x4*>(res_coef + 3 * (long)C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) rv[e] = edrl_bn_pre(rv[e], rm[e], rsc[e], rsf[e]);
      }
      v += rv;
    }
    if (relu) {
      if (mask_out) {   // 4 ReLU sign bits per float4: the backward reads this byte instead of the 16-B activation
        int mb = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) mb |= (v[e] > 0.f ? 1 : 0) << e;
        mask_out[i] = (unsigned char)mb;
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    st4<TA>(out + r * ld + c, v);
  }
}

// g = dout * (out>0);  dx = gamma*rstd*(g - c1 - xhat*c2);  dres (+)= g
template <typename TR, typename TA>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const TA* __restrict__ dout, const TA* __restrict__ out, const unsigned char* __restrict__ rmask,
    const TR* __restrict__ x,
    const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ gamma,
    const float* __restrict__ coef, TR* __restrict__ dx, TA* __restrict__ dres, int dres_accum, long M,
    int C, long ld) {
  const int C4 = C >> 2;
  const long total = M * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C4;
    const int c = (int)(i - r * C4) * 4;
    f32x4 g = ld4<TA>(dout + r * ld + c);
    if (rmask) {
      const int mb = rmask[i];
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = (mb >> e) & 1 ? g[e] : 0.f;
    } else if (out) {
      const f32x4 o = ld4<TA>(out + r * ld + c);
#pragma unroll
      for (int e = 0; e < 4; ++e) g[e] = o[e] > 0.f ? g[e] : 0.f;
    }
    const f32x4 xv = ld4<TR>(x + r * ld + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c);
    const f32x4 rs = *reinterpret_cast<const f32x4*>(rstd + c);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(coef + c);
    const f32x4 c2 = *reinterpret_cast<const f32x4*>(coef + C + c);
    f32x4 gm = {1.f, 1.f, 1.f, 1.f};
    if (gamma) gm = *reinterpret_cast<const f32x4*>(gamma + c);
    const f32x4 xh = (xv - mu) * rs;
    const f32x4 d = gm * rs * (g - c1 - xh * c2);
    if (dres) {
      f32x4 v = g;
      if (dres_accum) v += ld4<TA>(dres + r * ld + c);
      st4<TA>(dres + r * ld + c, v);
    }
    st4<TR>(dx + r * ld + c, d);
  }
}

// ------------------------------------------------------------------ pooling / layout
// 3x3 stride-2 pad-1 max pool on NHWC; idx = kh*3+kw of the first maximum (scan order kh, kw).
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               unsigned char* __restrict__ idx, int N, int H,
                                                               int W, int C, int Ho, int Wo) {
  const long total = (long)N * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    float best = -INFINITY;
    int bi = 0;
    bool any = false;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = ho * 2 - 1 + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = wo * 2 - 1 + kw;
        if (w < 0 || w >= W) continue;
        const float v = x[(((long)n * H + h) * W + w) * C + c];
        if (!any || v > best || v != v) { best = v; bi = kh * 3 + kw; any = true; }
      }
    }
    y[i] = best;
    idx[i] = (unsigned char)bi;
  }
}

__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_kernel(const float* __restrict__ dy,
                                                               const unsigned char* __restrict__ idx,
                                                               float* __restrict__ dx, int N, int H, int W, int C,
                                                               int Ho, int Wo) {
  const long total = (long)N * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    float s = 0.f;
    // windows (ho,wo) with ho*2-1+kh == h, kh in 0..2
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int th = h + 1 - kh;
      if (th < 0 || (th & 1)) continue;
      const int ho = th >> 1;
      if (ho >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tw = w + 1 - kw;
        if (tw < 0 || (tw & 1)) continue;
        const int wo = tw >> 1;
        if (wo >= Wo) continue;
        const long o = (((long)n * Ho + ho) * Wo + wo) * C + c;
        if (idx[o] == kh * 3 + kw) s += dy[o];
      }
    }
    dx[i] = s;
  }
}

// float4 / 4-byte-index variants (C % 4 == 0): 16 B per lane instead of 4
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_fwd_v4_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                  unsigned char* __restrict__ idx, int N, int H, int W,
                                                                  int C, int Ho, int Wo) {
  const int C4 = C >> 2;
  const long total = (long)N * Ho * Wo * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int wo = (int)(t % Wo); t /= Wo;
    const int ho = (int)(t % Ho);
    const int n = (int)(t / Ho);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool any = false;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = ho * 2 - 1 + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = wo * 2 - 1 + kw;
        if (w < 0 || w >= W) continue;
        const f32x4 v = ld4<T>(x + (((long)n * H + h) * W + w) * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (!any || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = kh * 3 + kw; }
        any = true;
      }
    }
    st4<T>(y + i * 4, best);
    *reinterpret_cast<unsigned int*>(idx + i * 4) = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
  }
}
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_bwd_v4_kernel(const T* __restrict__ dy,
                                                                  const unsigned char* __restrict__ idx,
                                                                  T* __restrict__ dx, int N, int H, int W, int C,
                                                                  int Ho, int Wo) {
  const int C4 = C >> 2;
  const long total = (long)N * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C4) * 4;
    long t = i / C4;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int th = h + 1 - kh;
      if (th < 0 || (th & 1)) continue;
      const int ho = th >> 1;
      if (ho >= Ho) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int tw = w + 1 - kw;
        if (tw < 0 || (tw & 1)) continue;
        const int wo = tw >> 1;
        if (wo >= Wo) continue;
        const long o = (((long)n * Ho + ho) * Wo + wo) * C + c;
        const unsigned m = *reinterpret_cast<const unsigned int*>(idx + o);
        const f32x4 g = ld4<T>(dy + o);
        const unsigned tap = (unsigned)(kh * 3 + kw);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (((m >> (8 * e)) & 0xff) == tap) s[e] += g[e];
      }
    }
    st4<T>(dx + i * 4, s);
  }
}

// ---- stem: BatchNorm + ReLU folded into the 3x3/s2 max-pool (the stem's activated tensor, 64 x 112^2 per 224^2 image, and its
// sign bytes are never stored).  Forward: y = max over the window of relu(x*scale + shift2) on the RAW stem conv output; idx as in
// maxpool3x3s2_fwd (first maximum in scan order).  Backward, two kernels that both rebuild the max-pool gradient on the fly
// (gather form: an input pixel collects dy of every window whose arg-max tap points at it) and the ReLU decision from the raw
// tensor: MODE 0 reduces (sum g, sum g*xhat) per 1024-pixel chunk [chunk][3][C] (colstat layout), MODE 1 writes
// d_raw = A*g + nK2*x + C2.  fcoef [5][C], bcoef [4][C] as in the fused conv kernels.
template <typename TY, typename TX = float>
__global__ __launch_bounds__(256) void maxpool_bn_fwd_kernel(const TX* __restrict__ x, const float* __restrict__ fcoef,
                                                             TY* __restrict__ y, unsigned char* __restrict__ idx, int N, int H,
                                                             int W, int C, int Ho, int Wo) {
  const int C4 = C >> 2;
  const long total = (long)N * Ho * Wo * C4;
  // the (n, ho, wo, c) of a thread's element: decoded once (the only 64-bit divisions), then advanced by the grid stride with
  // carries -- the stride is decomposed into (dn, dho, dwo, dc) up front
  const long gs = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int c4 = (int)(i % C4);
  long t0 = i / C4;
  int wo = (int)(t0 % Wo); t0 /= Wo;
  int ho = (int)(t0 % Ho);
  int n = (int)(t0 / Ho);
  const int dc = (int)(gs % C4);
  long t1 = gs / C4;
  const int dwo = (int)(t1 % Wo); t1 /= Wo;
  const int dho = (int)(t1 % Ho);
  const int dn = (int)(t1 / Ho);
  for (; i < total; i += gs) {
    const int c = c4 * 4;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(fcoef + 2 * (long)C + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(fcoef + 4 * (long)C + c);
    f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    int bi[4] = {0, 0, 0, 0};
    bool any = false;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = ho * 2 - 1 + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = wo * 2 - 1 + kw;
        if (w < 0 || w >= W) continue;
        const f32x4 v = edrl_bn_relu2(ld4<TX>(x + (((long)n * H + h) * W + w) * C + c), sc, sh);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (!any || v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = kh * 3 + kw; }
        any = true;
      }
    }
    st4<TY>(y + i * 4, best);
    *reinterpret_cast<unsigned int*>(idx + i * 4) = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
    c4 += dc; wo += dwo; ho += dho; n += dn;
    if (c4 >= C4) { c4 -= C4; ++wo; }
    if (wo >= Wo) { wo -= Wo; ++ho; }
    if (ho >= Ho) { ho -= Ho; ++n; }
  }
}

// Gradient of the fused stem (BatchNorm + ReLU + 3x3/s2/p1 max-pool) at input pixel (n, h, w), channels c..c+3: the windows that
// contain the pixel are enumerated directly -- an even h lies in ONE window row (ho = h/2, tap kh = 1), an odd h in two
// (ho = (h+1)/2 with kh = 0, ho = (h-1)/2 with kh = 2), likewise for w -- so at most 4 (dy, arg-max byte) pairs are read and no
// tap is tested that cannot match (the scan over all 9 taps with its parity tests was most of this kernel's instruction count).
template <typename TY>
__device__ __forceinline__ f32x4 maxpool_bn_gather_g(const TY* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                     const f32x4 xr, const f32x4 sc, const f32x4 sh, int n, int h, int w, int c,
                                                     int C, int Ho, int Wo) {
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  const int nh = (h & 1) ? 2 : 1, nw = (w & 1) ? 2 : 1;
  const int ho0 = (h + 1) >> 1, wo0 = (w + 1) >> 1;          // first candidate: kh = (h&1) ? 0 : 1
  const int kh0 = (h & 1) ? 0 : 1, kw0 = (w & 1) ? 0 : 1;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    if (a >= nh) break;
    const int ho = ho0 - a, kh = kh0 + 2 * a;                  // second candidate (odd h only): ho = (h-1)/2, kh = 2
    if (ho >= Ho) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (b >= nw) break;
      const int wo = wo0 - b, kw = kw0 + 2 * b;
      if (wo >= Wo) continue;
      const long o = (((long)n * Ho + ho) * Wo + wo) * C + c;
      const unsigned m = *reinterpret_cast<const unsigned int*>(idx + o);
      const f32x4 g = ld4<TY>(dy + o);
      const unsigned tap = (unsigned)(kh * 3 + kw);
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (((m >> (8 * e)) & 0xff) == tap) s[e] += g[e];
    }
  }
  const f32x4 pre = edrl_bn_pre2(xr, sc, sh);
#pragma unroll
  for (int e = 0; e < 4; ++e) s[e] = pre[e] > 0.f ? s[e] : 0.f;
  return s;
}

#define MPB_ROWS_PER_BLOCK 512     // MODE 1: pixels per workgroup
template <int MODE, typename TY, typename TX = float, typename TD = float>
__global__ __launch_bounds__(256) void maxpool_bn_bwd_kernel(const TY* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                             const TX* __restrict__ x, const float* __restrict__ fcoef,
                                                             const float* __restrict__ bcoef, float* __restrict__ part,
                                                             TD* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
  // Both modes: grid (row chunks, C/256 column blocks); a workgroup walks a CONTIGUOUS pixel range with CG = min(C/4, 64) column
  // lanes x RL = 256/CG pixel lanes; the (n, h, w) of a lane's pixel is decoded once (the only 64-bit division) and then advanced
  // by RL pixels per step with carries -- no per-element division.
  // MODE 0: chunks of BN_ROWS_PER_CHUNK pixels, reduction as colstat_kernel<1>;  MODE 1: MPB_ROWS_PER_BLOCK pixels, d_raw written.
  const int C4 = C >> 2;
  const long M = (long)N * H * W;
  const int CG = C4 < 64 ? C4 : 64;
  const int RL = 256 / CG;
  const int tid = threadIdx.x;
  const int cg = tid % CG, rl = tid / CG;
  const int c = (blockIdx.y * 64 + cg) * 4;
  const long row0 = (long)blockIdx.x * (MODE == 1 ? MPB_ROWS_PER_BLOCK : BN_ROWS_PER_CHUNK);
  long row1 = row0 + (MODE == 1 ? MPB_ROWS_PER_BLOCK : BN_ROWS_PER_CHUNK);
  if (row1 > M) row1 = M;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
  if (c < C && rl < RL && row0 + rl < row1) {
    const f32x4 sc = *reinterpret_cast<const f32x4*>(fcoef + 2 * (long)C + c);
    const f32x4 sh = *reinterpret_cast<const f32x4*>(fcoef + 4 * (long)C + c);
    f32x4 p0, p1, p2;
    if (MODE == 1) {
      p0 = *reinterpret_cast<const f32x4*>(bcoef + c);                 // A
      p1 = *reinterpret_cast<const f32x4*>(bcoef + (long)C + c);       // nK2
      p2 = *reinterpret_cast<const f32x4*>(bcoef + 2 * (long)C + c);   // C2
    } else {
      p0 = *reinterpret_cast<const f32x4*>(fcoef + c);                 // mean
      p1 = *reinterpret_cast<const f32x4*>(fcoef + (long)C + c);       // rstd
      p2 = p0;
    }
    long r = row0 + rl;
    int w = (int)(r % W);
    const long t = r / W;
    int h = (int)(t % H), n = (int)(t / H);
    for (; r < row1; r += RL) {
      const f32x4 xr = ld4<TX>(x + r * C + c);
      const f32x4 g = maxpool_bn_gather_g<TY>(dy, idx, xr, sc, sh, n, h, w, c, C, Ho, Wo);
      if (MODE == 1) {
        st4<TD>(dx + r * C + c, edrl_bn_bwd_dx2(g, xr, p0, p1, p2));
      } else {
        s0 += g;
        s1 += g * ((xr - p0) * p1);
      }
      w += RL;
      while (w >= W) { w -= W; if (++h == H) { h = 0; ++n; } }
    }
  }
  if (MODE == 1) return;
  __shared__ float sh_[256 * 8];
  float* my = sh_ + tid * 8;
#pragma unroll
  for (int e = 0; e < 4; ++e) { my[e] = s0[e]; my[4 + e] = s1[e]; }
  __syncthreads();
  if (rl == 0 && c < C) {
    float a[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) a[e] = 0.f;
    for (int q = 0; q < RL; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) a[e] += sh_[(q * CG + cg) * 8 + e];
    float* p = part + (long)blockIdx.x * 3 * C;
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[c + e] = a[e]; p[C + c + e] = a[4 + e]; }
  }
}

// ---- all-bf16 forms of the three fused stem kernels (raw tensor, pooled tensor / its gradient and d_raw stored as bf16: the bf16
// trunk), 8 channels per thread: the 4-channel kernels above move 8 bytes per load at bf16 and run instruction-bound (2.3 TB/s);
// with 16-byte accesses the instruction count per byte halves.  Same arithmetic per channel (results bit-identical to the 4-channel
// kernels: tests/test_gpu_bf16.py).  C % 8 == 0.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void ld8h(const __bf16* p, f32x4& lo, f32x4& hi) {
  const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(p);
#pragma unroll
  for (int e = 0; e < 4; ++e) { lo[e] = (float)v[e]; hi[e] = (float)v[4 + e]; }
}
__device__ __forceinline__ void st8h(__bf16* p, f32x4 lo, f32x4 hi) {
  bf16x8_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = (__bf16)lo[e]; o[4 + e] = (__bf16)hi[e]; }
  *reinterpret_cast<bf16x8_t*>(p) = o;
}
__global__ __launch_bounds__(256) void maxpool_bn_fwd8_kernel(const __bf16* __restrict__ x, const float* __restrict__ fcoef,
                                                              __bf16* __restrict__ y, unsigned char* __restrict__ idx, int N, int H,
                                                              int W, int C, int Ho, int Wo) {
  const int C8 = C >> 3;
  const long total = (long)N * Ho * Wo * C8;
  const long gs = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  int c8 = (int)(i % C8);
  long t0 = i / C8;
  int wo = (int)(t0 % Wo); t0 /= Wo;
  int ho = (int)(t0 % Ho);
  int n = (int)(t0 / Ho);
  const int dc = (int)(gs % C8);
  long t1 = gs / C8;
  const int dwo = (int)(t1 % Wo); t1 /= Wo;
  const int dho = (int)(t1 % Ho);
  const int dn = (int)(t1 / Ho);
  for (; i < total; i += gs) {
    const int c = c8 * 8;
    f32x4 sc[2], sh[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      sc[q] = *reinterpret_cast<const f32x4*>(fcoef + 2 * (long)C + c + 4 * q);
      sh[q] = *reinterpret_cast<const f32x4*>(fcoef + 4 * (long)C + c + 4 * q);
    }
    f32x4 best[2] = {{-INFINITY, -INFINITY, -INFINITY, -INFINITY}, {-INFINITY, -INFINITY, -INFINITY, -INFINITY}};
    int bi[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    bool any = false;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int h = ho * 2 - 1 + kh;
      if (h < 0 || h >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int w = wo * 2 - 1 + kw;
        if (w < 0 || w >= W) continue;
        f32x4 v[2];
        ld8h(x + (((long)n * H + h) * W + w) * C + c, v[0], v[1]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          v[q] = edrl_bn_relu2(v[q], sc[q], sh[q]);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            if (!any || v[q][e] > best[q][e] || v[q][e] != v[q][e]) { best[q][e] = v[q][e]; bi[4 * q + e] = kh * 3 + kw; }
        }
        any = true;
      }
    }
    st8h(y + i * 8, best[0], best[1]);
    uint2 pk;
    pk.x = (unsigned)bi[0] | ((unsigned)bi[1] << 8) | ((unsigned)bi[2] << 16) | ((unsigned)bi[3] << 24);
    pk.y = (unsigned)bi[4] | ((unsigned)bi[5] << 8) | ((unsigned)bi[6] << 16) | ((unsigned)bi[7] << 24);
    *reinterpret_cast<uint2*>(idx + i * 8) = pk;
    c8 += dc; wo += dwo; ho += dho; n += dn;
    if (c8 >= C8) { c8 -= C8; ++wo; }
    if (wo >= Wo) { wo -= Wo; ++ho; }
    if (ho >= Ho) { ho -= Ho; ++n; }
  }
}
// gradient gather of maxpool_bn_gather_g for 8 channels (two f32x4 halves)
__device__ __forceinline__ void maxpool_bn_gather_g8(const __bf16* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                     const f32x4 (&xr)[2], const f32x4 (&sc)[2], const f32x4 (&sh)[2], int n, int h,
                                                     int w, int c, int C, int Ho, int Wo, f32x4 (&s)[2]) {
  s[0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[1] = s[0];
  const int nh = (h & 1) ? 2 : 1, nw = (w & 1) ? 2 : 1;
  const int ho0 = (h + 1) >> 1, wo0 = (w + 1) >> 1;
  const int kh0 = (h & 1) ? 0 : 1, kw0 = (w & 1) ? 0 : 1;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    if (a >= nh) break;
    const int ho = ho0 - a, kh = kh0 + 2 * a;
    if (ho >= Ho) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      if (b >= nw) break;
      const int wo = wo0 - b, kw = kw0 + 2 * b;
      if (wo >= Wo) continue;
      const long o = (((long)n * Ho + ho) * Wo + wo) * C + c;
      const uint2 m = *reinterpret_cast<const uint2*>(idx + o);
      f32x4 g[2];
      ld8h(dy + o, g[0], g[1]);
      const unsigned tap = (unsigned)(kh * 3 + kw);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (((m.x >> (8 * e)) & 0xff) == tap) s[0][e] += g[0][e];
        if (((m.y >> (8 * e)) & 0xff) == tap) s[1][e] += g[1][e];
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const f32x4 pre = edrl_bn_pre2(xr[q], sc[q], sh[q]);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[q][e] = pre[e] > 0.f ? s[q][e] : 0.f;
  }
}
// Branch-free form of the gather (round 4): the (up to) four windows that cover an input pixel are all requested at once through
// buffer descriptors over dy / idx -- a window that does not exist for this pixel's parity (or lies past the edge) is an out-of-range
// offset, which reads as zeros and an index byte that matches no tap.  The loop form above predicates each window's loads on the
// pixel's parity: lanes of one wave differ in parity, so every wave ran all four iterations with one exposed latency each.
__device__ __forceinline__ void maxpool_bn_gather_g8_buf(const __amdgpu_buffer_rsrc_t rdy, const __amdgpu_buffer_rsrc_t ridx,
                                                         const f32x4 (&xr)[2], const f32x4 (&sc)[2], const f32x4 (&sh)[2], int n, int h,
                                                         int w, int c, int C, int Ho, int Wo, f32x4 (&s)[2]) {
  constexpr unsigned OOBP = 0x80000000u;
  const int nh = (h & 1) ? 2 : 1, nw = (w & 1) ? 2 : 1;
  const int ho0 = (h + 1) >> 1, wo0 = (w + 1) >> 1;
  const int kh0 = (h & 1) ? 0 : 1, kw0 = (w & 1) ? 0 : 1;
  bf16x8_t gv[4];
  unsigned mx[4], my[4];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ho = ho0 - a, wo = wo0 - b;
      const bool ok = a < nh && b < nw && ho < Ho && wo < Wo;
      const unsigned o = (unsigned)(((n * Ho + ho) * Wo + wo) * C + c);      // (element offset < 2^30: host-checked)
      const auto m2 = __builtin_amdgcn_raw_buffer_load_b64(ridx, ok ? (int)o : (int)OOBP, 0, 0);
      mx[2 * a + b] = ok ? (unsigned)m2[0] : 0xffffffffu;
      my[2 * a + b] = ok ? (unsigned)m2[1] : 0xffffffffu;
      gv[2 * a + b] = __builtin_bit_cast(bf16x8_t, __builtin_amdgcn_raw_buffer_load_b128(rdy, ok ? (int)(o * 2u) : (int)OOBP, 0, 0));
    }
  s[0] = f32x4{0.f, 0.f, 0.f, 0.f}; s[1] = s[0];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const unsigned tap = (unsigned)((kh0 + 2 * a) * 3 + kw0 + 2 * b);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (((mx[2 * a + b] >> (8 * e)) & 0xff) == tap) s[0][e] += (float)gv[2 * a + b][e];
        if (((my[2 * a + b] >> (8 * e)) & 0xff) == tap) s[1][e] += (float)gv[2 * a + b][4 + e];
      }
    }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const f32x4 pre = edrl_bn_pre2(xr[q], sc[q], sh[q]);
#pragma unroll
    for (int e = 0; e < 4; ++e) s[q][e] = pre[e] > 0.f ? s[q][e] : 0.f;
  }
}
template <int MODE>
__global__ __launch_bounds__(256) void maxpool_bn_bwd8_kernel(const __bf16* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                              const __bf16* __restrict__ x, const float* __restrict__ fcoef,
                                                              const float* __restrict__ bcoef, float* __restrict__ part,
                                                              __bf16* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo) {
  // as maxpool_bn_bwd_kernel with CG = min(C/8, 32) column lanes of 8 channels
  const int C8 = C >> 3;
  const long M = (long)N * H * W;
  const int CG = C8 < 32 ? C8 : 32;
  const int RL = 256 / CG;
  const int tid = threadIdx.x;
  const int cg = tid % CG, rl = tid / CG;
  const int c = (blockIdx.y * 32 + cg) * 8;
  const long row0 = (long)blockIdx.x * (MODE == 1 ? MPB_ROWS_PER_BLOCK : BN_ROWS_PER_CHUNK);
  long row1 = row0 + (MODE == 1 ? MPB_ROWS_PER_BLOCK : BN_ROWS_PER_CHUNK);
  if (row1 > M) row1 = M;
  f32x4 s0[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, s1[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  if (c < C && rl < RL && row0 + rl < row1) {
    f32x4 sc[2], sh[2], p0[2], p1[2], p2[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      sc[q] = *reinterpret_cast<const f32x4*>(fcoef + 2 * (long)C + c + 4 * q);
      sh[q] = *reinterpret_cast<const f32x4*>(fcoef + 4 * (long)C + c + 4 * q);
      if (MODE == 1) {
        p0[q] = *reinterpret_cast<const f32x4*>(bcoef + c + 4 * q);
        p1[q] = *reinterpret_cast<const f32x4*>(bcoef + (long)C + c + 4 * q);
        p2[q] = *reinterpret_cast<const f32x4*>(bcoef + 2 * (long)C + c + 4 * q);
      } else {
        p0[q] = *reinterpret_cast<const f32x4*>(fcoef + c + 4 * q);
        p1[q] = *reinterpret_cast<const f32x4*>(fcoef + (long)C + c + 4 * q);
        p2[q] = p0[q];
      }
    }
    long r = row0 + rl;
    int w = (int)(r % W);
    const long t = r / W;
    int h = (int)(t % H), n = (int)(t / H);
    // pooled tensors below 2^30 elements: the four windows of a pixel through descriptors (see maxpool_bn_gather_g8_buf)
    const long pooled = (long)N * Ho * Wo * C;
    const bool usebuf = pooled < (1L << 30);
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc((void*)dy, 0, usebuf ? (int)(pooled * 2) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t ridx = __builtin_amdgcn_make_buffer_rsrc((void*)idx, 0, usebuf ? (int)pooled : 0, 0x00020000);
    for (; r < row1; r += RL) {
      f32x4 xr[2], g[2];
      ld8h(x + r * C + c, xr[0], xr[1]);
      if (usebuf) maxpool_bn_gather_g8_buf(rdy, ridx, xr, sc, sh, n, h, w, c, C, Ho, Wo, g);
      else maxpool_bn_gather_g8(dy, idx, xr, sc, sh, n, h, w, c, C, Ho, Wo, g);
      if (MODE == 1) {
        st8h(dx + r * C + c, edrl_bn_bwd_dx2(g[0], xr[0], p0[0], p1[0], p2[0]), edrl_bn_bwd_dx2(g[1], xr[1], p0[1], p1[1], p2[1]));
      } else {
#pragma unroll
        for (int q = 0; q < 2; ++q) { s0[q] += g[q]; s1[q] += g[q] * ((xr[q] - p0[q]) * p1[q]); }
      }
      w += RL;
      while (w >= W) { w -= W; if (++h == H) { h = 0; ++n; } }
    }
  }
  if (MODE == 1) return;
  __shared__ float sh_[256 * 16];
  float* my = sh_ + tid * 16;
#pragma unroll
  for (int e = 0; e < 4; ++e) { my[e] = s0[0][e]; my[4 + e] = s0[1][e]; my[8 + e] = s1[0][e]; my[12 + e] = s1[1][e]; }
  __syncthreads();
  if (rl == 0 && c < C) {
    float a[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.f;
    for (int q = 0; q < RL; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) a[e] += sh_[(q * CG + cg) * 16 + e];
    float* p = part + (long)blockIdx.x * 3 * C;
#pragma unroll
    for (int e = 0; e < 8; ++e) { p[c + e] = a[e]; p[C + c + e] = a[8 + e]; }
  }
}

// [N][C][H][W] -> [N][H][W][Cp] (channels >= C zero filled)
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                           int N, int C, int H, int W, int Cp) {
  const long total = (long)N * H * W * Cp;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Cp);
    long t = i / Cp;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H);
    const int n = (int)(t / H);
    out[i] = c < C ? in[(((long)n * C + c) * H + h) * W + w] : 0.f;
  }
}

// in [A][L][D] -> out [A][D] = scale * sum_l in
__global__ __launch_bounds__(256) void sum_axis1_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                        long A, int L, int D, float scale) {
  const long total = A * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long a = i / D;
    const int d = (int)(i - a * D);
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += in[(a * L + l) * D + d];
    out[i] = s * scale;
  }
}
// float4 columns, the L rows of one `a` spread over RL = TPB/CG row lanes (fixed order -> deterministic): the per-image global
// average pool (A = B*S images, L = 49) and the bias gradients (A = 1, L = all rows) no longer walk L serially per column.
template <int TPB>
__global__ __launch_bounds__(TPB) void sum_axis1_v4_kernel(const float* __restrict__ in, float* __restrict__ out, long A,
                                                           int L, int D, float scale) {
  __shared__ f32x4 sh[TPB];
  const int C4 = D >> 2;
  const int CG = C4 < 64 ? C4 : 64;
  const int RL = TPB / CG;
  const int tid = threadIdx.x, cg = tid % CG, rl = tid / CG;
  const int c4 = blockIdx.x * CG + cg;
  const long a = blockIdx.y;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (c4 < C4 && rl < RL) {
    const float* p = in + (a * L) * D + (long)c4 * 4;
    int l = rl;
    for (; l + RL < L; l += 2 * RL) {
      s0 += *reinterpret_cast<const f32x4*>(p + (long)l * D);
      s1 += *reinterpret_cast<const f32x4*>(p + (long)(l + RL) * D);
    }
    if (l < L) s0 += *reinterpret_cast<const f32x4*>(p + (long)l * D);
  }
  sh[tid] = s0 + s1;
  __syncthreads();
  if (rl == 0 && c4 < C4) {
    f32x4 t = sh[cg];
    for (int q = 1; q < RL; ++q) t += sh[q * CG + cg];
    *reinterpret_cast<f32x4*>(out + a * D + (long)c4 * 4) = t * scale;
  }
}
// out [A][L][D] (+)= scale * in [A][D]
__global__ __launch_bounds__(256) void bcast_axis1_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                          long A, int L, int D, float scale, int accumulate) {
  const long total = A * L * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const long a = i / ((long)L * D);
    const float v = in[a * D + d] * scale;
    out[i] = accumulate ? out[i] + v : v;
  }
}

// bn_apply_kernel: one workgroup per 1024-float4 chunk, up to 64 workgroups per CU in the queue (the chunked fast path loops
// beyond that)
static inline int bn_apply_grid(long total) {
  long b = (total + 1023) / 1024;
  if (b > 256 * 64) b = 256 * 64;
  if (b < 1) b = 1;
  return (int)b;
}
// colstat1_h8_kernel: 8-channel groups per workgroup (64 -> 512 channels; 16 when the grid would be under ~4 workgroups per CU)
static inline int colstat_h8_groups(long chunks, int C) {
  return (chunks * ((C + 511) / 512) < 1024 && C >= 256) ? 16 : 64;
}
static inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// ---- stem re-layout: a 7x7 / stride-2 / pad-3 convolution over C channels equals a 4x4 / stride-1 convolution over the
// 2x2 space-to-depth image with 4C channels (top/left pad 2, bottom/right pad 1; the 8th tap per axis has zero weight):
//   input row i = 2*oh - 3 + kh = 2*(oh - 2 + ka) + ph   with kh + 1 = 2*ka + ph.
// For the single-channel OCT stem this turns K = 49 scalar gathers into K = 64 with one 16-byte load per tap (the four
// phases of a 2x2 block are the four channels), i.e. the vector MFMA path instead of the scalar fallback.
// y[n,a,b,(ph*2+pw)*C + c] = x[n,2a+ph,2b+pw,c]
__global__ __launch_bounds__(256) void s2d_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const int H2 = H >> 1, W2 = W >> 1;
  const long total = (long)N * H2 * W2 * 4 * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ph = (int)((t >> 1) & 1), pw = (int)(t & 1);
    t >>= 2;
    const int b = (int)(t % W2);
    t /= W2;
    const int a = (int)(t % H2);
    const int n = (int)(t / H2);
    y[i] = x[(((long)n * H + 2 * a + ph) * W + 2 * b + pw) * C + c];
  }
}
// dir 0: w7 [Co,7,7,C] -> w8 [Co,4,4,4C] (zero taps filled);  dir 1: w8 -> w7 (gather; used on the weight gradient)
__global__ __launch_bounds__(256) void stem_weight_fold_kernel(const float* __restrict__ in, float* __restrict__ out, int Co,
                                                               int C, int dir) {
  const long total = dir == 0 ? (long)Co * 16 * 4 * C : (long)Co * 49 * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (dir == 0) {
      const int c = (int)(i % C);
      long t = i / C;
      const int ph = (int)((t >> 1) & 1), pw = (int)(t & 1);
      t >>= 2;
      const int kb = (int)(t & 3), ka = (int)((t >> 2) & 3);
      const int co = (int)(t >> 4);
      const int kh = 2 * ka + ph - 1, kw = 2 * kb + pw - 1;
      out[i] = (kh >= 0 && kw >= 0) ? in[(((long)co * 7 + kh) * 7 + kw) * C + c] : 0.f;
    } else {
      const int c = (int)(i % C);
      long t = i / C;
      const int kw = (int)(t % 7);
      t /= 7;
      const int kh = (int)(t % 7);
      const int co = (int)(t / 7);
      const int ka = (kh + 1) >> 1, ph = (kh + 1) & 1, kb = (kw + 1) >> 1, pw = (kw + 1) & 1;
      out[i] = in[((((long)co * 4 + ka) * 4 + kb) * 4 + (ph * 2 + pw)) * C + c];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 / bf16 specialisations, 8 channels (one 16-byte access) per lane: the templated kernels above move 8 bytes per
// lane on bf16 tensors, which leaves HBM bandwidth on the table (measured 4.1 vs 5.0 TB/s).  C % 8 == 0.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void ld8h(const __bf16* p, float* o) {
  const bf16x8 v = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (float)v[e];
}
__device__ __forceinline__ void st8h(__bf16* p, const float* o) {
  bf16x8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = (__bf16)o[e];
  *reinterpret_cast<bf16x8*>(p) = v;
}
__device__ __forceinline__ void ld8f(const float* p, float* o) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
}

// Both h8 elementwise kernels: when C/8 is a power of two <= 256 (every ResNet width) a thread keeps ONE column group, so the
// per-channel coefficients are loaded once, the row index is a shift, and a workgroup streams block-contiguous chunks of 4 x 256
// 16-byte groups with all of a thread's loads in flight before the first use (the shape that reaches ~6 TB/s here,
// scripts/hbm_probe2.py); the arithmetic is unchanged (bit-identical results).  Other widths take the grid-stride loop.
__global__ __launch_bounds__(256) void bn_apply_h8_kernel(const __bf16* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const __bf16* __restrict__ residual, __bf16* __restrict__ out,
                                                          unsigned char* __restrict__ mask_out, long M, int C, int relu,
                                                          const float* __restrict__ res_coef = nullptr) {
  const int C8 = C >> 3;
  const long total = M * C8;
  auto one = [&](long i, long r, int c, const float* xv, float* rv, const float* mu, const float* sc, const float* sf,
                 const float* rm, const float* rsc, const float* rsf) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (xv[e] - mu[e]) * sc[e] + sf[e];
    if (residual) {
      if (res_coef) {      // the residual is the raw downsample-branch output: its BatchNorm is applied here
#pragma unroll
        for (int e = 0; e < 8; ++e) rv[e] = (rv[e] - rm[e]) * rsc[e] + rsf[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] += rv[e];
    }
    if (relu) {
      if (mask_out) {   // same layout as the 4-wide kernels: one byte (4 sign bits) per 4 channels -> two bytes here
        int mb = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) mb |= (v[e] > 0.f ? 1 : 0) << (e < 4 ? e : e + 4);
        *reinterpret_cast<unsigned short*>(mask_out + i * 2) = (unsigned short)mb;
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    st8h(out + r * C + c, v);
  };
  if ((C8 & (C8 - 1)) == 0 && C8 <= 256) {
    const int lg = __builtin_ctz(C8);
    const int c = (int)(threadIdx.x & (C8 - 1)) * 8;
    float mu[8], sc[8], sf[8], rm[8], rsc[8], rsf[8];
    ld8f(mean + c, mu); ld8f(scale + c, sc); ld8f(shift + c, sf);
    if (residual && res_coef) { ld8f(res_coef + c, rm); ld8f(res_coef + 2 * (long)C + c, rsc); ld8f(res_coef + 3 * (long)C + c, rsf); }
    const long cstride = (long)gridDim.x * 1024;
    for (long base = (long)blockIdx.x * 1024; base < total; base += cstride) {
      bf16x8 xb[4], rb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = base + u * 256 + threadIdx.x;
        if (i < total) {
          xb[u] = *reinterpret_cast<const bf16x8*>(x + (i >> lg) * C + c);
          if (residual) rb[u] = *reinterpret_cast<const bf16x8*>(residual + (i >> lg) * C + c);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = base + u * 256 + threadIdx.x;
        if (i < total) {
          float xv[8], rv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { xv[e] = (float)xb[u][e]; rv[e] = residual ? (float)rb[u][e] : 0.f; }
          one(i, i >> lg, c, xv, rv, mu, sc, sf, rm, rsc, rsf);
        }
      }
    }
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C8;
    const int c = (int)(i - r * C8) * 8;
    float xv[8], mu[8], sc[8], sf[8], rv[8], rm[8], rsc[8], rsf[8];
    ld8h(x + r * C + c, xv);
    ld8f(mean + c, mu); ld8f(scale + c, sc); ld8f(shift + c, sf);
    if (residual) {
      ld8h(residual + r * C + c, rv);
      if (res_coef) { ld8f(res_coef + c, rm); ld8f(res_coef + 2 * (long)C + c, rsc); ld8f(res_coef + 3 * (long)C + c, rsf); }
    }
    one(i, r, c, xv, rv, mu, sc, sf, rm, rsc, rsf);
  }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_h8_kernel(const __bf16* __restrict__ dout,
                                                              const unsigned char* __restrict__ rmask,
                                                              const __bf16* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ coef, __bf16* __restrict__ dx,
                                                              __bf16* __restrict__ dres, long M, int C) {
  const int C8 = C >> 3;
  const long total = M * C8;
  auto one = [&](long i, long r, int c, float* g, const float* xv, int mb, const float* mu, const float* rs, const float* c1,
                 const float* c2, const float* gm) {
    float d[8];
    if (rmask) {
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = (mb >> (e < 4 ? e : e + 4)) & 1 ? g[e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float xh = (xv[e] - mu[e]) * rs[e];
      d[e] = (gamma ? gm[e] : 1.f) * rs[e] * (g[e] - c1[e] - xh * c2[e]);
    }
    if (dres) st8h(dres + r * C + c, g);
    st8h(dx + r * C + c, d);
  };
  if ((C8 & (C8 - 1)) == 0 && C8 <= 256) {
    const int lg = __builtin_ctz(C8);
    const int c = (int)(threadIdx.x & (C8 - 1)) * 8;
    float mu[8], rs[8], c1[8], c2[8], gm[8];
    ld8f(mean + c, mu); ld8f(rstd + c, rs); ld8f(coef + c, c1); ld8f(coef + C + c, c2);
    if (gamma) ld8f(gamma + c, gm);
    const long cstride = (long)gridDim.x * 1024;
    for (long base = (long)blockIdx.x * 1024; base < total; base += cstride) {
      bf16x8 gb[4], xb[4];
      int mbs[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = base + u * 256 + threadIdx.x;
        mbs[u] = 0;
        if (i < total) {
          gb[u] = *reinterpret_cast<const bf16x8*>(dout + (i >> lg) * C + c);
          xb[u] = *reinterpret_cast<const bf16x8*>(x + (i >> lg) * C + c);
          if (rmask) mbs[u] = *reinterpret_cast<const unsigned short*>(rmask + i * 2);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = base + u * 256 + threadIdx.x;
        if (i < total) {
          float g[8], xv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { g[e] = (float)gb[u][e]; xv[e] = (float)xb[u][e]; }
          one(i, i >> lg, c, g, xv, mbs[u], mu, rs, c1, c2, gm);
        }
      }
    }
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C8;
    const int c = (int)(i - r * C8) * 8;
    float g[8], xv[8], mu[8], rs[8], c1[8], c2[8], gm[8];
    ld8h(dout + r * C + c, g);
    ld8h(x + r * C + c, xv);
    const int mb = rmask ? *reinterpret_cast<const unsigned short*>(rmask + i * 2) : 0;
    ld8f(mean + c, mu); ld8f(rstd + c, rs); ld8f(coef + c, c1); ld8f(coef + C + c, c2);
    if (gamma) ld8f(gamma + c, gm);
    one(i, r, c, g, xv, mb, mu, rs, c1, c2, gm);
  }
}

// d_raw = A*g + nK2*x + C2 (bcoef rows 0..2) as a standalone bf16 pass: what the fused conv kernels form in their operand loads,
// materialised for a consumer that runs on the plain kernels (the 3x3 layer of a fused bf16 bottleneck block, encoders._KBF16.mid_sep).
__global__ __launch_bounds__(256) void bn_draw_h8_kernel(const __bf16* __restrict__ g, const __bf16* __restrict__ x,
                                                         const float* __restrict__ bcoef, __bf16* __restrict__ out, long M, int C) {
  const int C8 = C >> 3;
  const long total = M * C8;
  auto one = [&](long r, int c, const bf16x8 gb, const bf16x8 xb, const float* A, const float* nK2, const float* C2) {
    float d[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) d[e] = __builtin_fmaf(nK2[e], (float)xb[e], __builtin_fmaf(A[e], (float)gb[e], C2[e]));
    st8h(out + r * C + c, d);
  };
  if ((C8 & (C8 - 1)) == 0 && C8 <= 256) {
    const int lg = __builtin_ctz(C8);
    const int c = (int)(threadIdx.x & (C8 - 1)) * 8;
    float A[8], nK2[8], C2[8];
    ld8f(bcoef + c, A); ld8f(bcoef + (long)C + c, nK2); ld8f(bcoef + 2 * (long)C + c, C2);
    const long cstride = (long)gridDim.x * 1024;
    for (long base = (long)blockIdx.x * 1024; base < total; base += cstride) {
      bf16x8 gb[4], xb[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = base + u * 256 + threadIdx.x;
        if (i < total) {
          gb[u] = *reinterpret_cast<const bf16x8*>(g + (i >> lg) * C + c);
          xb[u] = *reinterpret_cast<const bf16x8*>(x + (i >> lg) * C + c);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const long i = base + u * 256 + threadIdx.x;
        if (i < total) one(i >> lg, c, gb[u], xb[u], A, nK2, C2);
      }
    }
    return;
  }
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / C8;
    const int c = (int)(i - r * C8) * 8;
    float A[8], nK2[8], C2[8];
    ld8f(bcoef + c, A); ld8f(bcoef + (long)C + c, nK2); ld8f(bcoef + 2 * (long)C + c, C2);
    one(r, c, *reinterpret_cast<const bf16x8*>(g + r * C + c), *reinterpret_cast<const bf16x8*>(x + r * C + c), A, nK2, C2);
  }
}

// (sum g, sum g*xhat) partials, g = dout * relu-mask: part[chunk][3][C] like colstat_kernel<1>
__global__ __launch_bounds__(256) void colstat1_h8_kernel(const __bf16* __restrict__ x, const __bf16* __restrict__ dout,
                                                          const unsigned char* __restrict__ rmask,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          long M, int C, float* __restrict__ part,
                                                          __bf16* __restrict__ gout = nullptr, int cgmax = 64) {
  __shared__ float sh[256 * 16];
  const int C8 = C >> 3;
  // 8-channel groups per block: 64 (512 channels) on big grids; 16 when the (chunks x C/512) grid would leave most CUs without
  // enough waves (stage 3-4 tensors: ~100-400 chunks) -- 4x the workgroups, 16 rows of the chunk in parallel per workgroup
  const int CG = C8 < cgmax ? C8 : cgmax;
  const int RL = 256 / CG;
  const int tid = threadIdx.x;
  const int cg = tid % CG, rl = tid / CG;
  const int c = (blockIdx.y * cgmax + cg) * 8;
  const long row0 = (long)blockIdx.x * BN_ROWS_PER_CHUNK;
  long row1 = row0 + BN_ROWS_PER_CHUNK;
  if (row1 > M) row1 = M;
  float s0[8], s1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
  if (c < C && rl < RL) {
    float mu[8], rs[8];
    ld8f(mean + c, mu); ld8f(rstd + c, rs);
    constexpr int U = 4;          // rows in flight per thread (the running sums are still taken in row order)
    long r = row0 + rl;
    auto acc = [&](const float* xv, float* g, int mb, bool masked) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ge = (!masked || ((mb >> (e < 4 ? e : e + 4)) & 1)) ? g[e] : 0.f;
        g[e] = ge;
        s0[e] += ge;
        s1[e] += ge * ((xv[e] - mu[e]) * rs[e]);
      }
    };
    for (; r + (long)(U - 1) * RL < row1; r += (long)U * RL) {
      float xv[U][8], gv[U][8];
      int mb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long rr = r + (long)u * RL;
        ld8h(x + rr * C + c, xv[u]);
        ld8h(dout + rr * C + c, gv[u]);
        mb[u] = rmask ? *reinterpret_cast<const unsigned short*>(rmask + (rr * C8 + (c >> 3)) * 2) : 0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        acc(xv[u], gv[u], mb[u], rmask != nullptr);
        if (gout) st8h(gout + (r + (long)u * RL) * C + c, gv[u]);
      }
    }
    for (; r < row1; r += RL) {
      float xv[8], gv[8];
      ld8h(x + r * C + c, xv);
      ld8h(dout + r * C + c, gv);
      const int mb = rmask ? *reinterpret_cast<const unsigned short*>(rmask + (r * C8 + (c >> 3)) * 2) : 0;
      acc(xv, gv, mb, rmask != nullptr);
      if (gout) st8h(gout + r * C + c, gv);
    }
  }
  float* my = sh + tid * 16;
#pragma unroll
  for (int e = 0; e < 8; ++e) { my[e] = s0[e]; my[8 + e] = s1[e]; }
  __syncthreads();
  if (rl == 0 && c < C) {
    float a[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.f;
    for (int q = 0; q < RL; ++q)
#pragma unroll
      for (int e = 0; e < 16; ++e) a[e] += sh[(q * CG + cg) * 16 + e];
    float* p = part + (long)blockIdx.x * 3 * C;
#pragma unroll
    for (int e = 0; e < 8; ++e) { p[c + e] = a[e]; p[C + c + e] = a[8 + e]; }
  }
}

extern "C" {

size_t edrl_bn_workspace_bytes(long M, int C) {
  const long chunks = (M + BN_ROWS_PER_CHUNK - 1) / BN_ROWS_PER_CHUNK;
  return (size_t)chunks * 3 * C * sizeof(float);
}

// Train-mode batch statistics of x [M][C] (row stride ld) and the affine that applies them.
int edrl_bn_train_stats_f32(const float* x, long M, int C, long ld, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps,
                            float* save_mean, float* save_rstd, float* scale, float* shift, float* workspace,
                            size_t workspace_bytes, hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 3) || (ld & 3) || ld < C) return EDRL_EINVAL;
  if (workspace_bytes < edrl_bn_workspace_bytes(M, C)) return EDRL_ENOSPC;
  const int chunks = edrl_cdiv(M, BN_ROWS_PER_CHUNK);
  hipLaunchKernelGGL((colstat_kernel<0, float, float>), dim3(chunks, edrl_cdiv(C, 256)), dim3(256), 0, st, x,
                     (const float*)nullptr, (const float*)nullptr, (const unsigned char*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, M, C, ld, workspace);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(edrl_cdiv(C, FIN_CH)), dim3(256), 0, st, workspace, chunks, C, M,
                     BN_ROWS_PER_CHUNK, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd, scale,
                     shift);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// Second half of edrl_bn_train_stats_f32 for chunk partials produced elsewhere (the conv epilogue,
// edrl_conv2d_nhwc_fwd_stats_f32): part [nchunks][3][C], chunk k covering rows [k*rows_per_chunk, ...).
size_t edrl_bn_finalize_group_ws_bytes(long nchunks, int C) {
  return (size_t)((nchunks + FIN_GROUP - 1) / FIN_GROUP) * 3 * C * sizeof(double);
}
int edrl_bn_finalize_partials_f32(const float* part, long nchunks, int rows_per_chunk, long M, int C, const float* gamma,
                                  const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                  float* save_mean, float* save_rstd, float* scale, float* shift, double* group_ws,
                                  size_t group_ws_bytes, hipStream_t st) {
  if (M <= 0 || C <= 0 || nchunks <= 0 || nchunks > 0x7fffffffL || rows_per_chunk <= 0 ||
      nchunks != (M + rows_per_chunk - 1) / rows_per_chunk)
    return EDRL_EINVAL;
  if (group_ws && nchunks > 2 * FIN_GROUP) {
    if (group_ws_bytes < edrl_bn_finalize_group_ws_bytes(nchunks, C)) return EDRL_ENOSPC;
    const int G = (int)((nchunks + FIN_GROUP - 1) / FIN_GROUP);
    hipLaunchKernelGGL(bn_group_kernel, dim3(edrl_cdiv(C, FIN_CH), G), dim3(256), 0, st, part, (int)nchunks, C, M,
                       rows_per_chunk, group_ws);
    EDRL_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_groups_kernel, dim3(edrl_cdiv(C, FIN_CH)), dim3(256), 0, st, group_ws, G, C, M, gamma, beta,
                       running_mean, running_var, momentum, eps, save_mean, save_rstd, scale, shift);
    EDRL_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(edrl_cdiv(C, FIN_CH)), dim3(256), 0, st, part, (int)nchunks, C, M,
                     rows_per_chunk, gamma, beta, running_mean, running_var, momentum, eps, save_mean, save_rstd, scale,
                     shift);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// Same reduction, output as ONE coefficient array fcoef [5][C] = {mean, rstd, scale, shift, shift2 = shift - mean*scale}
// (the layout the fused conv kernels and edrl_bn_apply_res_f32 / edrl_bn_bwd_reduce_f32 take).
int edrl_bn_finalize_fcoef_f32(const float* part, long nchunks, int rows_per_chunk, long M, int C, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                               float* fcoef, double* group_ws, size_t group_ws_bytes, hipStream_t st) {
  if (M <= 0 || C <= 0 || nchunks <= 0 || nchunks > 0x7fffffffL || rows_per_chunk <= 0 || !fcoef ||
      nchunks != (M + rows_per_chunk - 1) / rows_per_chunk)
    return EDRL_EINVAL;
  float *mean = fcoef, *rstd = fcoef + C, *scale = fcoef + 2 * (long)C, *shift = fcoef + 3 * (long)C, *shift2 = fcoef + 4 * (long)C;
  if (group_ws && nchunks > 2 * FIN_GROUP) {
    if (group_ws_bytes < edrl_bn_finalize_group_ws_bytes(nchunks, C)) return EDRL_ENOSPC;
    const int G = (int)((nchunks + FIN_GROUP - 1) / FIN_GROUP);
    hipLaunchKernelGGL(bn_group_kernel, dim3(edrl_cdiv(C, FIN_CH), G), dim3(256), 0, st, part, (int)nchunks, C, M,
                       rows_per_chunk, group_ws);
    EDRL_LAUNCH_CHECK();
    hipLaunchKernelGGL(bn_finalize_groups_kernel, dim3(edrl_cdiv(C, FIN_CH)), dim3(256), 0, st, group_ws, G, C, M, gamma, beta,
                       running_mean, running_var, momentum, eps, mean, rstd, scale, shift, shift2);
    EDRL_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(edrl_cdiv(C, FIN_CH)), dim3(256), 0, st, part, (int)nchunks, C, M,
                     rows_per_chunk, gamma, beta, running_mean, running_var, momentum, eps, mean, rstd, scale, shift, shift2);
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_bn_apply_f32(const float* x, const float* mean, const float* scale, const float* shift,
                      const float* residual, float* out, unsigned char* relu_mask, long M, int C, long ld, int relu,
                      hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 3) || (ld & 3)) return EDRL_EINVAL;
  if (relu_mask && ld != C) return EDRL_EINVAL;   // the byte mask is dense [M][C/4]
  hipLaunchKernelGGL((bn_apply_kernel<float, float>), dim3(bn_apply_grid(M * (C / 4))), dim3(256), 0, st, x, mean, scale, shift, residual,
                     out, relu_mask, M, C, ld, relu);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// BN(+ReLU)(+residual) backward.  dout: grad of the post-activation output; out: that output (NULL = no ReLU).
// dx: grad of the raw (pre-BN) tensor; dres (optional): grad of the residual operand, (+)= dout*mask.
int edrl_bn_bwd_f32(const float* dout, const float* out, const unsigned char* relu_mask, const float* x,
                    const float* save_mean, const float* save_rstd, const float* gamma, float* dgamma, float* dbeta,
                    int accumulate, float* dx, float* dres, int dres_accum, long M, int C, long ld, float* workspace,
                    size_t workspace_bytes, hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 3) || (ld & 3)) return EDRL_EINVAL;
  if (relu_mask && ld != C) return EDRL_EINVAL;
  const size_t stats = edrl_bn_workspace_bytes(M, C);
  if (workspace_bytes < stats + (size_t)2 * C * sizeof(float)) return EDRL_ENOSPC;
  float* coef = workspace + stats / sizeof(float);
  const int chunks = edrl_cdiv(M, BN_ROWS_PER_CHUNK);
  hipLaunchKernelGGL((colstat_kernel<1, float, float>), dim3(chunks, edrl_cdiv(C, 256)), dim3(256), 0, st, x, dout, out, relu_mask,
                     save_mean, save_rstd, M, C, ld, workspace);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(edrl_cdiv(C, FIN_CH)), dim3(256), 0, st, workspace, chunks, C, M,
                     dgamma, dbeta, accumulate, coef);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL((bn_bwd_apply_kernel<float, float>), dim3(ew_grid(M * (C / 4))), dim3(256), 0, st, dout, out, relu_mask, x,
                     save_mean, save_rstd, gamma, coef, dx, dres, dres_accum, M, C, ld);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// BatchNorm apply whose residual operand may itself be a raw conv output with its own BatchNorm (res_fcoef [5][C] =
// {mean, rstd, scale, shift}; NULL: the residual is used as is):  out = act((x-mean)*scale+shift + bn_r(residual)).
int edrl_bn_apply_res_f32(const float* x, const float* fcoef, const float* residual, const float* res_fcoef, float* out,
                          unsigned char* relu_mask, long M, int C, int relu, hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 3) || !x || !fcoef || !out) return EDRL_EINVAL;
  hipLaunchKernelGGL((bn_apply_kernel<float, float>), dim3(bn_apply_grid(M * (C / 4))), dim3(256), 0, st, x, fcoef,
                     fcoef + 2 * (long)C, fcoef + 3 * (long)C, residual, out, relu_mask, M, C, (long)C, relu, res_fcoef);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// First half of a BatchNorm(+ReLU) backward as a standalone pass (the trunk's last block, whose upstream gradient comes from
// autograd, and the downsample branch): g = dout * relu-mask (mask bytes, optional) is written to g_out (optional) and the
// partial sums (sum g, sum g*xhat) go to part [ceil(M/1024)][3][C] (planes = 3 for edrl_bn_bwd_finalize_partials_f32).
int edrl_bn_bwd_reduce_f32(const float* dout, const unsigned char* relu_mask, const float* x, const float* fcoef, float* g_out,
                           float* part, size_t part_bytes, long M, int C, hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 3) || !dout || !x || !fcoef || !part) return EDRL_EINVAL;
  if (part_bytes < edrl_bn_workspace_bytes(M, C)) return EDRL_ENOSPC;
  const int chunks = edrl_cdiv(M, BN_ROWS_PER_CHUNK);
  hipLaunchKernelGGL((colstat_kernel<1, float, float>), dim3(chunks, edrl_cdiv(C, 256)), dim3(256), 0, st, x, dout,
                     (const float*)nullptr, relu_mask, fcoef, fcoef + (long)C, M, C, (long)C, part, g_out);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// Second half: partial sums -> dgamma, dbeta and the coefficients bcoef [4][C] = {A, nK2, C2, mean} from which the fused conv
// kernels form d_raw = A*g - K1 - K2*(x - mean) in their operand loads.  group_ws: nchunks/64 x 2 x C doubles.
size_t edrl_bn_bwd_group_ws_bytes(long nchunks, int C) {
  return (size_t)((nchunks + FIN_GROUP - 1) / FIN_GROUP) * 2 * C * sizeof(double);
}
int edrl_bn_bwd_finalize_partials_f32(const float* part, long nchunks, int planes, long M, int C, const float* gamma,
                                      const float* fcoef, float* dgamma, float* dbeta, float* bcoef, double* group_ws,
                                      size_t group_ws_bytes, hipStream_t st) {
  if (M <= 0 || C <= 0 || nchunks <= 0 || nchunks > 0x7fffffffL || (planes != 2 && planes != 3) || !part || !fcoef || !bcoef)
    return EDRL_EINVAL;
  const double* groups = nullptr;
  int G = 0;
  if (nchunks > 2 * FIN_GROUP) {
    if (!group_ws || group_ws_bytes < edrl_bn_bwd_group_ws_bytes(nchunks, C)) return EDRL_ENOSPC;
    G = (int)((nchunks + FIN_GROUP - 1) / FIN_GROUP);
    hipLaunchKernelGGL(bn_bwd_group_kernel, dim3(edrl_cdiv(C, FIN_CH), G), dim3(256), 0, st, part, (int)nchunks, planes, C,
                       group_ws);
    EDRL_LAUNCH_CHECK();
    groups = group_ws;
  }
  hipLaunchKernelGGL(bn_bwd_finalize_coef_kernel, dim3(edrl_cdiv(C, FIN_CH)), dim3(256), 0, st, part, (int)nchunks, planes,
                     groups, G, C, M, gamma, fcoef, fcoef + (long)C, dgamma, dbeta, bcoef);
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_space_to_depth2_f32(const float* x, float* y, int N, int H, int W, int C, hipStream_t st) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1)) return EDRL_EINVAL;
  hipLaunchKernelGGL(s2d_kernel, dim3(ew_grid((long)N * H * W * C)), dim3(256), 0, st, x, y, N, H, W, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_stem_weight_fold_f32(const float* in, float* out, int Co, int C, int dir, hipStream_t st) {
  if (Co <= 0 || C <= 0 || (dir != 0 && dir != 1)) return EDRL_EINVAL;
  hipLaunchKernelGGL(stem_weight_fold_kernel, dim3(ew_grid((long)Co * 64 * C)), dim3(256), 0, st, in, out, Co, C, dir);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_maxpool3x3s2_fwd_f32(const float* x, float* y, unsigned char* idx, int N, int H, int W, int C,
                              hipStream_t st) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return EDRL_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if ((C & 3) == 0 && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)idx) & 15) == 0)
    hipLaunchKernelGGL(maxpool3x3s2_fwd_v4_kernel<float>, dim3(ew_grid((long)N * Ho * Wo * (C / 4))), dim3(256), 0, st, x, y, idx,
                       N, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool3x3s2_fwd_kernel, dim3(ew_grid((long)N * Ho * Wo * C)), dim3(256), 0, st, x, y, idx, N,
                       H, W, C, Ho, Wo);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_maxpool3x3s2_bwd_f32(const float* dy, const unsigned char* idx, float* dx, int N, int H, int W, int C,
                              hipStream_t st) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0) return EDRL_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if ((C & 3) == 0 && (((uintptr_t)dy | (uintptr_t)dx | (uintptr_t)idx) & 15) == 0)
    hipLaunchKernelGGL(maxpool3x3s2_bwd_v4_kernel<float>, dim3(ew_grid((long)N * H * W * (C / 4))), dim3(256), 0, st, dy, idx, dx,
                       N, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool3x3s2_bwd_kernel, dim3(ew_grid((long)N * H * W * C)), dim3(256), 0, st, dy, idx, dx, N,
                       H, W, C, Ho, Wo);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// Stem: BatchNorm(train) statistics of x [M][C] as ONE coefficient array fcoef [5][C] (edrl_bn_train_stats_f32 otherwise).
int edrl_bn_train_stats_fcoef_f32(const float* x, long M, int C, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float momentum, float eps, float* fcoef, float* workspace,
                                  size_t workspace_bytes, hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 3) || !fcoef) return EDRL_EINVAL;
  if (workspace_bytes < edrl_bn_workspace_bytes(M, C)) return EDRL_ENOSPC;
  const int chunks = edrl_cdiv(M, BN_ROWS_PER_CHUNK);
  hipLaunchKernelGGL((colstat_kernel<0, float, float>), dim3(chunks, edrl_cdiv(C, 256)), dim3(256), 0, st, x,
                     (const float*)nullptr, (const float*)nullptr, (const unsigned char*)nullptr, (const float*)nullptr,
                     (const float*)nullptr, M, C, (long)C, workspace);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(edrl_cdiv(C, FIN_CH)), dim3(256), 0, st, workspace, chunks, C, M,
                     BN_ROWS_PER_CHUNK, gamma, beta, running_mean, running_var, momentum, eps, fcoef, fcoef + C,
                     fcoef + 2 * (long)C, fcoef + 3 * (long)C, fcoef + 4 * (long)C);
  EDRL_LAUNCH_CHECK();
  return 0;
}
// Stem max-pool with the BatchNorm + ReLU of its input folded in (x = RAW stem conv output [N,H,W,C], fcoef [5][C]):
// y [N,Ho,Wo,C] = maxpool3x3/s2/p1(relu(x*scale + shift2)), idx = arg-max tap bytes.
// _mx: x_bf16 = 1: the raw stem output is a bf16 tensor (the bf16 trunk's stem, edrl_conv2d_nhwc_fwd_stats_f32_obf16; needs
// y_bf16 = 1); y_bf16 = 1: the pooled tensor is bf16.
int edrl_maxpool3x3s2_bn_fwd_mx(const void* x, int x_bf16, const float* fcoef, void* y, int y_bf16, unsigned char* idx, int N, int H,
                                int W, int C, hipStream_t st) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || !fcoef || (x_bf16 && !y_bf16)) return EDRL_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const dim3 grid(ew_grid((long)N * Ho * Wo * (C / 4)));
  const bool v8_env = edrl_cfg().stem_pool_v8 != 0;     // (A/B switch EDRL_STEM_POOL_V8)
  if (x_bf16 && v8_env && (C & 7) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0 && (((uintptr_t)idx) & 7) == 0)
    hipLaunchKernelGGL(maxpool_bn_fwd8_kernel, dim3(ew_grid((long)N * Ho * Wo * (C / 8))), dim3(256), 0, st, (const __bf16*)x, fcoef,
                       (__bf16*)y, idx, N, H, W, C, Ho, Wo);
  else if (x_bf16)
    hipLaunchKernelGGL((maxpool_bn_fwd_kernel<__bf16, __bf16>), grid, dim3(256), 0, st, (const __bf16*)x, fcoef, (__bf16*)y, idx, N, H, W,
                       C, Ho, Wo);
  else if (y_bf16)
    hipLaunchKernelGGL(maxpool_bn_fwd_kernel<__bf16>, grid, dim3(256), 0, st, (const float*)x, fcoef, (__bf16*)y, idx, N, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL(maxpool_bn_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, fcoef, (float*)y, idx, N, H, W, C, Ho, Wo);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_maxpool3x3s2_bn_fwd_f32(const float* x, const float* fcoef, float* y, unsigned char* idx, int N, int H, int W, int C,
                                 hipStream_t st) {
  return edrl_maxpool3x3s2_bn_fwd_mx(x, 0, fcoef, y, 0, idx, N, H, W, C, st);
}
// Its backward in two launches (the max-pool gradient and the ReLU decision are rebuilt on the fly in both):
//   _reduce: partial sums (sum g, sum g*xhat) -> part [ceil(N*H*W/1024)][3][C] (planes = 3 for edrl_bn_bwd_finalize_partials_f32)
//   _apply : d_raw [N,H,W,C] = A*g + nK2*x + C2 with bcoef [4][C]
// _mx: the pooled tensor's gradient dy is bf16 (dy_bf16 = 1, the bf16 trunk) or fp32; the raw stem output x is bf16 with
// x_bf16 = 1 (needs dy_bf16 = 1); d_raw stays fp32 (it feeds the fp32 stem weight gradient).
int edrl_maxpool3x3s2_bn_bwd_reduce_mx(const void* dy, int dy_bf16, const unsigned char* idx, const void* x, int x_bf16,
                                       const float* fcoef, float* part, size_t part_bytes, int N, int H, int W, int C,
                                       hipStream_t st) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || !fcoef || !part || (x_bf16 && !dy_bf16)) return EDRL_EINVAL;
  const long M = (long)N * H * W;
  if (part_bytes < edrl_bn_workspace_bytes(M, C)) return EDRL_ENOSPC;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const dim3 grid(edrl_cdiv(M, BN_ROWS_PER_CHUNK), edrl_cdiv(C, 256));
  const bool v8_env = edrl_cfg().stem_pool_v8 != 0;     // (A/B switch EDRL_STEM_POOL_V8)
  if (x_bf16 && v8_env && (C & 7) == 0 && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0 && (((uintptr_t)idx) & 7) == 0)
    hipLaunchKernelGGL((maxpool_bn_bwd8_kernel<0>), dim3(edrl_cdiv(M, BN_ROWS_PER_CHUNK), edrl_cdiv(C, 256)), dim3(256), 0, st,
                       (const __bf16*)dy, idx, (const __bf16*)x, fcoef, (const float*)nullptr, part, (__bf16*)nullptr, N, H, W, C, Ho, Wo);
  else if (x_bf16)
    hipLaunchKernelGGL((maxpool_bn_bwd_kernel<0, __bf16, __bf16>), grid, dim3(256), 0, st, (const __bf16*)dy, idx, (const __bf16*)x,
                       fcoef, (const float*)nullptr, part, (float*)nullptr, N, H, W, C, Ho, Wo);
  else if (dy_bf16)
    hipLaunchKernelGGL((maxpool_bn_bwd_kernel<0, __bf16>), grid, dim3(256), 0, st, (const __bf16*)dy, idx, (const float*)x, fcoef,
                       (const float*)nullptr, part, (float*)nullptr, N, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL((maxpool_bn_bwd_kernel<0, float>), grid, dim3(256), 0, st, (const float*)dy, idx, (const float*)x, fcoef,
                       (const float*)nullptr, part, (float*)nullptr, N, H, W, C, Ho, Wo);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_maxpool3x3s2_bn_bwd_reduce_f32(const float* dy, const unsigned char* idx, const float* x, const float* fcoef, float* part,
                                        size_t part_bytes, int N, int H, int W, int C, hipStream_t st) {
  return edrl_maxpool3x3s2_bn_bwd_reduce_mx(dy, 0, idx, x, 0, fcoef, part, part_bytes, N, H, W, C, st);
}
int edrl_maxpool3x3s2_bn_bwd_apply_mx(const void* dy, int dy_bf16, const unsigned char* idx, const void* x, int x_bf16,
                                      const float* fcoef, const float* bcoef, void* d_raw, int d_bf16, int N, int H, int W, int C,
                                      hipStream_t st) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || !fcoef || !bcoef || !d_raw || (x_bf16 && !dy_bf16) || (d_bf16 && !x_bf16))
    return EDRL_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const dim3 grid(edrl_cdiv((long)N * H * W, MPB_ROWS_PER_BLOCK), edrl_cdiv(C, 256));
  const bool v8_env = edrl_cfg().stem_pool_v8 != 0;     // (A/B switch EDRL_STEM_POOL_V8)
  if (d_bf16 && v8_env && (C & 7) == 0 && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)d_raw) & 15) == 0 && (((uintptr_t)idx) & 7) == 0)
    hipLaunchKernelGGL((maxpool_bn_bwd8_kernel<1>), grid, dim3(256), 0, st, (const __bf16*)dy, idx, (const __bf16*)x, fcoef, bcoef,
                       (float*)nullptr, (__bf16*)d_raw, N, H, W, C, Ho, Wo);
  else if (d_bf16)
    hipLaunchKernelGGL((maxpool_bn_bwd_kernel<1, __bf16, __bf16, __bf16>), grid, dim3(256), 0, st, (const __bf16*)dy, idx, (const __bf16*)x,
                       fcoef, bcoef, (float*)nullptr, (__bf16*)d_raw, N, H, W, C, Ho, Wo);
  else if (x_bf16)
    hipLaunchKernelGGL((maxpool_bn_bwd_kernel<1, __bf16, __bf16>), grid, dim3(256), 0, st, (const __bf16*)dy, idx, (const __bf16*)x,
                       fcoef, bcoef, (float*)nullptr, (float*)d_raw, N, H, W, C, Ho, Wo);
  else if (dy_bf16)
    hipLaunchKernelGGL((maxpool_bn_bwd_kernel<1, __bf16>), grid, dim3(256), 0, st, (const __bf16*)dy, idx, (const float*)x, fcoef, bcoef,
                       (float*)nullptr, (float*)d_raw, N, H, W, C, Ho, Wo);
  else
    hipLaunchKernelGGL((maxpool_bn_bwd_kernel<1, float>), grid, dim3(256), 0, st, (const float*)dy, idx, (const float*)x, fcoef, bcoef,
                       (float*)nullptr, (float*)d_raw, N, H, W, C, Ho, Wo);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_maxpool3x3s2_bn_bwd_apply_f32(const float* dy, const unsigned char* idx, const float* x, const float* fcoef,
                                       const float* bcoef, float* d_raw, int N, int H, int W, int C, hipStream_t st) {
  return edrl_maxpool3x3s2_bn_bwd_apply_mx(dy, 0, idx, x, 0, fcoef, bcoef, d_raw, 0, N, H, W, C, st);
}

int edrl_nchw_to_nhwc_f32(const float* in, float* out, int N, int C, int H, int W, int Cp, hipStream_t st) {
  if (N <= 0 || C <= 0 || H <= 0 || W <= 0 || Cp < C) return EDRL_EINVAL;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(ew_grid((long)N * H * W * Cp)), dim3(256), 0, st, in, out, N, C, H,
                     W, Cp);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// out[a][d] = scale * sum_l in[a][l][d]   (global avg-pool, token mean)
int edrl_sum_axis1_f32(const float* in, float* out, long A, int L, int D, float scale, hipStream_t st) {
  if (A <= 0 || L <= 0 || D <= 0) return EDRL_EINVAL;
  if ((D & 3) == 0 && ((((uintptr_t)in) | ((uintptr_t)out)) & 15) == 0 && A <= 65535) {
    const int C4 = D >> 2, CG = C4 < 64 ? C4 : 64;
    const dim3 grid(edrl_cdiv(C4, CG), (unsigned)A);
    if ((long)grid.x * grid.y < 512 && L >= 64)      // few blocks, long columns (bias gradients): 16 row lanes per column group
      hipLaunchKernelGGL(sum_axis1_v4_kernel<1024>, grid, dim3(1024), 0, st, in, out, A, L, D, scale);
    else
      hipLaunchKernelGGL(sum_axis1_v4_kernel<256>, grid, dim3(256), 0, st, in, out, A, L, D, scale);
  } else {
    hipLaunchKernelGGL(sum_axis1_kernel, dim3(ew_grid(A * D)), dim3(256), 0, st, in, out, A, L, D, scale);
  }
  EDRL_LAUNCH_CHECK();
  return 0;
}
// out[a][l][d] (+)= scale * in[a][d]
int edrl_bcast_axis1_f32(const float* in, float* out, long A, int L, int D, float scale, int accumulate,
                         hipStream_t st) {
  if (A <= 0 || L <= 0 || D <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(bcast_axis1_kernel, dim3(ew_grid(A * L * D)), dim3(256), 0, st, in, out, A, L, D, scale,
                     accumulate);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// ---- mixed-precision variants (bf16 activations / gradients, fp32 arithmetic): raw_bf16 / act_bf16 select the types of
// the raw conv output (and its gradient) and of the activated tensors (and their gradients).  Dense rows (ld == C).
int edrl_bn_apply_mx(const void* x, int raw_bf16, const float* mean, const float* scale, const float* shift,
                     const void* residual, void* out, int act_bf16, unsigned char* relu_mask, long M, int C, int relu,
                     hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 3)) return EDRL_EINVAL;
  const dim3 grid(bn_apply_grid(M * (C / 4)));
  if (raw_bf16 && act_bf16 && (C & 7) == 0)
    hipLaunchKernelGGL(bn_apply_h8_kernel, dim3(bn_apply_grid(M * (C / 8))), dim3(256), 0, st, (const __bf16*)x, mean, scale, shift,
                       (const __bf16*)residual, (__bf16*)out, relu_mask, M, C, relu);
  else if (raw_bf16 && act_bf16)
    hipLaunchKernelGGL((bn_apply_kernel<__bf16, __bf16>), grid, dim3(256), 0, st, (const __bf16*)x, mean, scale, shift,
                       (const __bf16*)residual, (__bf16*)out, relu_mask, M, C, (long)C, relu);
  else if (!raw_bf16 && act_bf16)
    hipLaunchKernelGGL((bn_apply_kernel<float, __bf16>), grid, dim3(256), 0, st, (const float*)x, mean, scale, shift,
                       (const __bf16*)residual, (__bf16*)out, relu_mask, M, C, (long)C, relu);
  else if (!raw_bf16 && !act_bf16)
    hipLaunchKernelGGL((bn_apply_kernel<float, float>), grid, dim3(256), 0, st, (const float*)x, mean, scale, shift,
                       (const float*)residual, (float*)out, relu_mask, M, C, (long)C, relu);
  else
    return EDRL_EINVAL;
  EDRL_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

template <typename TR, typename TA>
static int bn_bwd_mx_impl(const TA* dout, const unsigned char* relu_mask, const TR* x, const float* save_mean,
                          const float* save_rstd, const float* gamma, float* dgamma, float* dbeta, TR* dx, TA* dres, long M,
                          int C, float* workspace, hipStream_t st) {
  const size_t stats = edrl_bn_workspace_bytes(M, C);
  float* coef = workspace + stats / sizeof(float);
  const int chunks = edrl_cdiv(M, BN_ROWS_PER_CHUNK);
  hipLaunchKernelGGL((colstat_kernel<1, TR, TA>), dim3(chunks, edrl_cdiv(C, 256)), dim3(256), 0, st, x, dout,
                     (const TA*)nullptr, relu_mask, save_mean, save_rstd, M, C, (long)C, workspace);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(edrl_cdiv(C, FIN_CH)), dim3(256), 0, st, workspace, chunks, C, M, dgamma,
                     dbeta, 0, coef);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL((bn_bwd_apply_kernel<TR, TA>), dim3(ew_grid(M * (C / 4))), dim3(256), 0, st, dout, (const TA*)nullptr,
                     relu_mask, x, save_mean, save_rstd, gamma, coef, dx, dres, 0, M, C, (long)C);
  EDRL_LAUNCH_CHECK();
  return 0;
}
static int bn_bwd_h8_impl(const __bf16* dout, const unsigned char* relu_mask, const __bf16* x, const float* save_mean,
                          const float* save_rstd, const float* gamma, float* dgamma, float* dbeta, __bf16* dx, __bf16* dres,
                          long M, int C, float* workspace, hipStream_t st) {
  const size_t stats = edrl_bn_workspace_bytes(M, C);
  float* coef = workspace + stats / sizeof(float);
  const int chunks = edrl_cdiv(M, BN_ROWS_PER_CHUNK);
  {
    const int cgm = colstat_h8_groups(chunks, C);
    hipLaunchKernelGGL(colstat1_h8_kernel, dim3(chunks, edrl_cdiv(C, cgm * 8)), dim3(256), 0, st, x, dout, relu_mask, save_mean,
                       save_rstd, M, C, workspace, (__bf16*)nullptr, cgm);
  }
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(edrl_cdiv(C, FIN_CH)), dim3(256), 0, st, workspace, chunks, C, M, dgamma,
                     dbeta, 0, coef);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(bn_bwd_apply_h8_kernel, dim3(bn_apply_grid(M * (C / 8))), dim3(256), 0, st, dout, relu_mask, x, save_mean,
                     save_rstd, gamma, coef, dx, dres, M, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}
extern "C" {

int edrl_bn_bwd_mx(const void* dout, int act_bf16, const unsigned char* relu_mask, const void* x, int raw_bf16,
                   const float* save_mean, const float* save_rstd, const float* gamma, float* dgamma, float* dbeta, void* dx,
                   void* dres, long M, int C, float* workspace, size_t workspace_bytes, hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 3)) return EDRL_EINVAL;
  if (workspace_bytes < edrl_bn_workspace_bytes(M, C) + (size_t)2 * C * sizeof(float)) return EDRL_ENOSPC;
  if (raw_bf16 && act_bf16 && (C & 7) == 0)
    return bn_bwd_h8_impl((const __bf16*)dout, relu_mask, (const __bf16*)x, save_mean, save_rstd, gamma, dgamma, dbeta,
                          (__bf16*)dx, (__bf16*)dres, M, C, workspace, st);
  if (raw_bf16 && act_bf16)
    return bn_bwd_mx_impl<__bf16, __bf16>((const __bf16*)dout, relu_mask, (const __bf16*)x, save_mean, save_rstd, gamma,
                                          dgamma, dbeta, (__bf16*)dx, (__bf16*)dres, M, C, workspace, st);
  if (!raw_bf16 && act_bf16)
    return bn_bwd_mx_impl<float, __bf16>((const __bf16*)dout, relu_mask, (const float*)x, save_mean, save_rstd, gamma, dgamma,
                                         dbeta, (float*)dx, (__bf16*)dres, M, C, workspace, st);
  return EDRL_EINVAL;
}
// bf16 counterparts of edrl_bn_apply_res_f32 / edrl_bn_bwd_reduce_f32 (bf16 raw tensors, activations and gradients; fp32
// coefficients and sums).  C % 8 == 0.
int edrl_bn_apply_res_bf16(const void* x, const float* fcoef, const void* residual, const float* res_fcoef, void* out,
                           unsigned char* relu_mask, long M, int C, int relu, hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 7) || !x || !fcoef || !out) return EDRL_EINVAL;
  hipLaunchKernelGGL(bn_apply_h8_kernel, dim3(bn_apply_grid(M * (C / 8))), dim3(256), 0, st, (const __bf16*)x, fcoef,
                     fcoef + 2 * (long)C, fcoef + 3 * (long)C, (const __bf16*)residual, (__bf16*)out, relu_mask, M, C, relu,
                     res_fcoef);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_bn_bwd_reduce_bf16(const void* dout, const unsigned char* relu_mask, const void* x, const float* fcoef, void* g_out,
                            float* part, size_t part_bytes, long M, int C, hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 7) || !dout || !x || !fcoef || !part) return EDRL_EINVAL;
  if (part_bytes < edrl_bn_workspace_bytes(M, C)) return EDRL_ENOSPC;
  const int chunks = edrl_cdiv(M, BN_ROWS_PER_CHUNK);
  {
    const int cgm = colstat_h8_groups(chunks, C);
    hipLaunchKernelGGL(colstat1_h8_kernel, dim3(chunks, edrl_cdiv(C, cgm * 8)), dim3(256), 0, st, (const __bf16*)x, (const __bf16*)dout,
                       relu_mask, fcoef, fcoef + (long)C, M, C, part, (__bf16*)g_out, cgm);
  }
  EDRL_LAUNCH_CHECK();
  return 0;
}

// the same pass over fp32 tensors (float4 per lane, four in flight): the fp32 trunk's plain-kernel units (encoders._K32.mid_sep / wide)
__global__ __launch_bounds__(256) void bn_draw_f4_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                         const float* __restrict__ bcoef, float* __restrict__ out, long M, int C) {
  const int C4 = C >> 2;
  const long total = M * C4;
  const long cstride = (long)gridDim.x * 1024;
  for (long base = (long)blockIdx.x * 1024; base < total; base += cstride) {
    f32x4 gv[4], xv[4];
    long idx[4];
    int cc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long i = base + u * 256 + threadIdx.x;
      idx[u] = i;
      if (i < total) {
        const long r = i / C4;
        cc[u] = (int)(i - r * C4) * 4;
        gv[u] = *reinterpret_cast<const f32x4*>(g + r * C + cc[u]);
        xv[u] = *reinterpret_cast<const f32x4*>(x + r * C + cc[u]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (idx[u] < total) {
        const f32x4 A = *reinterpret_cast<const f32x4*>(bcoef + cc[u]);
        const f32x4 nK2 = *reinterpret_cast<const f32x4*>(bcoef + (long)C + cc[u]);
        const f32x4 C2 = *reinterpret_cast<const f32x4*>(bcoef + 2 * (long)C + cc[u]);
        f32x4 d;
#pragma unroll
        for (int e = 0; e < 4; ++e) d[e] = __builtin_fmaf(nK2[e], xv[u][e], __builtin_fmaf(A[e], gv[u][e], C2[e]));
        *reinterpret_cast<f32x4*>(out + idx[u] * 4) = d;
      }
    }
  }
}

// d_raw = A*g + nK2*x + C2 with bcoef [4][C]; g, x, d_raw dense [M][C] fp32, C % 4 == 0: the same fma order as the operand
// transform of the fused conv kernels (edrl_bn_bwd_dx2 in conv_gemm.hip), so a unit gives the same d_raw either way.
int edrl_bn_draw_f32(const float* g, const float* x, const float* bcoef, float* d_raw, long M, int C, hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 3) || !g || !x || !bcoef || !d_raw) return EDRL_EINVAL;
  hipLaunchKernelGGL(bn_draw_f4_kernel, dim3(bn_apply_grid(M * (C / 4))), dim3(256), 0, st, g, x, bcoef, d_raw, M, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// d_raw (bf16) = A*g + nK2*x + C2 with bcoef [4][C] (edrl_bn_bwd_finalize_partials_f32); g, x, d_raw dense [M][C] bf16, C % 8 == 0.
int edrl_bn_draw_bf16(const void* g, const void* x, const float* bcoef, void* d_raw, long M, int C, hipStream_t st) {
  if (M <= 0 || C <= 0 || (C & 7) || !g || !x || !bcoef || !d_raw) return EDRL_EINVAL;
  hipLaunchKernelGGL(bn_draw_h8_kernel, dim3(bn_apply_grid(M * (C / 8))), dim3(256), 0, st, (const __bf16*)g, (const __bf16*)x, bcoef,
                     (__bf16*)d_raw, M, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_maxpool3x3s2_fwd_bf16(const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, hipStream_t st) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return EDRL_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_fwd_v4_kernel<__bf16>, dim3(ew_grid((long)N * Ho * Wo * (C / 4))), dim3(256), 0, st,
                     (const __bf16*)x, (__bf16*)y, idx, N, H, W, C, Ho, Wo);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_maxpool3x3s2_bwd_bf16(const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C,
                               hipStream_t st) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return EDRL_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  hipLaunchKernelGGL(maxpool3x3s2_bwd_v4_kernel<__bf16>, dim3(ew_grid((long)N * H * W * (C / 4))), dim3(256), 0, st,
                     (const __bf16*)dy, idx, (__bf16*)dx, N, H, W, C, Ho, Wo);
  EDRL_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
