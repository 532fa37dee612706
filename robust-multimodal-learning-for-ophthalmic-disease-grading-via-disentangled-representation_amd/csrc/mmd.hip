// Multi-kernel RBF MMD (code/MMD.py:3-74) — the fused reduction stage.
//
// The Gram matrix G = total @ total^T ([n][n], n = n_s + n_t) comes from the MFMA GEMM
// (edrl_conv2d_nhwc_fwd_f32 in its 1x1 form); `sq` = row sums of squares (MMD.py:25).
// This file fuses everything after it into one pass-structured single-workgroup kernel:
//   L2 = clamp(sq_i + sq_j - 2 G_ij, 0)                      MMD.py:26-27
//   bw = sum(L2)/(n^2-n) / mul^(num//2)                       MMD.py:31-34
//   K  = sum_q exp(-L2 / (bw * mul^q))                        MMD.py:37-42
//   loss = | mean(XX) + mean(YY) - mean(XY) - mean(YX) |      MMD.py:66-72
// and its backward (gradient through the data-dependent bandwidth included), which emits the
// [n][n] coefficient matrix Coef with dTotal = Coef @ total (one more MFMA GEMM).
// n <= 2048; the [n][n] tile is L2-cache resident, so one 1024-thread workgroup suffices.
#include "edrl_common.h"

__device__ __forceinline__ float block_sum_1024(float v, float* sh) {
  v = edrl_wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += sh[i];
  return s;
}

// rowsq[i] = sum_d x[i][d]^2, one wave per row
__global__ __launch_bounds__(256) void rowsq_kernel(const float* __restrict__ x, float* __restrict__ sq, int n, int d,
                                                    long ld) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < d; i += 64) { const float v = x[row * ld + i]; s += v * v; }
  s = edrl_wave_sum(s);
  if (lane == 0) sq[row] = s;
}

// saved[0] = bandwidth, saved[1] = signed sum (XX+YY-XY-YX), saved[2] = loss
__global__ __launch_bounds__(1024) void mmd_fwd_kernel(const float* __restrict__ G, const float* __restrict__ sq, int n,
                                                       int ns, float mul, int num, float* __restrict__ loss,
                                                       float* __restrict__ saved) {
  __shared__ float sh[16];
  const long total = (long)n * n;
  const int nt = n - ns;
  float s = 0.f;
  for (long i = threadIdx.x; i < total; i += 1024) {
    const int r = (int)(i / n), c = (int)(i - (long)r * n);
    s += fmaxf(sq[r] + sq[c] - 2.f * G[i], 0.f);
  }
  const float S = block_sum_1024(s, sh);
  float bw = S / (float)((long)n * n - n);
  bw /= powf(mul, (float)(num / 2));
  // Four quadrant sums, each reduced in the same thread/lane order: identical quadrants (MK_MMD(a,a))
  // give bit-identical sums and an exactly-zero loss, like the reference's four .sum() calls.
  float qs[4];
#pragma unroll
  for (int qd = 0; qd < 4; ++qd) {
    const int r0 = (qd & 2) ? ns : 0, c0 = (qd & 1) ? ns : 0;
    const int nr = (qd & 2) ? nt : ns, nc = (qd & 1) ? nt : ns;
    const long cnt = (long)nr * nc;
    float acc = 0.f;
    for (long i = threadIdx.x; i < cnt; i += 1024) {
      const int r = r0 + (int)(i / nc), c = c0 + (int)(i % nc);
      const float L = fmaxf(sq[r] + sq[c] - 2.f * G[(long)r * n + c], 0.f);
      float k = 0.f, beta = bw;
      for (int q = 0; q < num; ++q) { k += expf(-L / beta); beta *= mul; }
      acc += k;
    }
    qs[qd] = block_sum_1024(acc, sh);
  }
  const float XX = qs[0] / ((float)ns * (float)ns), YY = qs[3] / ((float)nt * (float)nt);
  const float XY = qs[1] / ((float)ns * (float)nt), YX = qs[2] / ((float)ns * (float)nt);
  const float tot = XX + YY - XY - YX;
  if (threadIdx.x == 0) {
    saved[0] = bw;
    saved[1] = tot;
    saved[2] = fabsf(tot);
    loss[0] = fabsf(tot);
  }
}

// Coef[i][j] = -2 (E_ij + E_ji) + delta_ij * 2 * sum_j (E_ij + E_ji),  E = dLoss/dL2 (clamp-masked).
// E is staged in `Ebuf` ([n][n] workspace).
__global__ __launch_bounds__(1024) void mmd_bwd_kernel(const float* __restrict__ dloss, const float* __restrict__ G,
                                                       const float* __restrict__ sq, const float* __restrict__ saved,
                                                       int n, int ns, float mul, int num, float* __restrict__ Ebuf,
                                                       float* __restrict__ coef) {
  __shared__ float sh[16];
  const long total = (long)n * n;
  const int nt = n - ns;
  const float bw = saved[0];
  const float sgn = saved[1] > 0.f ? 1.f : (saved[1] < 0.f ? -1.f : 0.f);
  const float gl = dloss[0] * sgn;
  const float wxx = 1.f / ((float)ns * (float)ns), wyy = 1.f / ((float)nt * (float)nt),
              wxy = 1.f / ((float)ns * (float)nt);
  // pass 1: direct term into Ebuf, accumulate d(bw)
  float dbw = 0.f;
  for (long i = threadIdx.x; i < total; i += 1024) {
    const int r = (int)(i / n), c = (int)(i - (long)r * n);
    const float L = fmaxf(sq[r] + sq[c] - 2.f * G[i], 0.f);
    const bool rs = r < ns, cs = c < ns;
    const float dK = gl * ((rs && cs) ? wxx : ((!rs && !cs) ? wyy : -wxy));
    float a = 0.f, t = 0.f, beta = bw, mq = 1.f;
    for (int q = 0; q < num; ++q) {
      const float e = expf(-L / beta);
      a += e * (-1.f / beta);
      t += e * (L / (beta * beta)) * mq;
      beta *= mul; mq *= mul;
    }
    Ebuf[i] = dK * a;
    dbw += dK * t;
  }
  dbw = block_sum_1024(dbw, sh);
  const float dS = dbw / ((float)((long)n * n - n) * powf(mul, (float)(num / 2)));
  __syncthreads();
  // pass 2: add bandwidth term, apply the clamp mask (torch.clamp passes grad where raw >= 0)
  for (long i = threadIdx.x; i < total; i += 1024) {
    const int r = (int)(i / n), c = (int)(i - (long)r * n);
    const float raw = sq[r] + sq[c] - 2.f * G[i];
    Ebuf[i] = raw >= 0.f ? Ebuf[i] + dS : 0.f;
  }
  __syncthreads();
  __threadfence_block();
  // pass 3: coefficient matrix; one wave per row for the row sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int r = wave; r < n; r += 16) {
    float rsum = 0.f;
    for (int c = lane; c < n; c += 64) rsum += Ebuf[(long)r * n + c] + Ebuf[(long)c * n + r];
    rsum = edrl_wave_sum(rsum);
    for (int c = lane; c < n; c += 64) {
      float v = -2.f * (Ebuf[(long)r * n + c] + Ebuf[(long)c * n + r]);
      if (c == r) v += 2.f * rsum;
      coef[(long)r * n + c] = v;
    }
  }
}

extern "C" {

int edrl_rowsq_f32(const float* x, float* sq, int n, int d, long ld, hipStream_t st) {
  if (n <= 0 || d <= 0 || ld < d) return EDRL_EINVAL;
  hipLaunchKernelGGL(rowsq_kernel, dim3(edrl_cdiv(n, 4)), dim3(256), 0, st, x, sq, n, d, ld);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// G [n][n], sq [n]; loss [1]; saved [3]
int edrl_mk_mmd_fwd_f32(const float* G, const float* sq, int n, int ns, float kernel_mul, int kernel_num, float* loss,
                        float* saved, hipStream_t st) {
  if (n < 2 || ns <= 0 || ns >= n || kernel_num <= 0 || n > 2048) return EDRL_EINVAL;
  hipLaunchKernelGGL(mmd_fwd_kernel, dim3(1), dim3(1024), 0, st, G, sq, n, ns, kernel_mul, kernel_num, loss, saved);
  EDRL_LAUNCH_CHECK();
  return 0;
}
// workspace: n*n floats; coef: n*n floats
int edrl_mk_mmd_bwd_f32(const float* dloss, const float* G, const float* sq, const float* saved, int n, int ns,
                        float kernel_mul, int kernel_num, float* workspace, float* coef, hipStream_t st) {
  if (n < 2 || ns <= 0 || ns >= n || kernel_num <= 0 || n > 2048) return EDRL_EINVAL;
  hipLaunchKernelGGL(mmd_bwd_kernel, dim3(1), dim3(1024), 0, st, dloss, G, sq, saved, n, ns, kernel_mul, kernel_num,
                     workspace, coef);
  EDRL_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
