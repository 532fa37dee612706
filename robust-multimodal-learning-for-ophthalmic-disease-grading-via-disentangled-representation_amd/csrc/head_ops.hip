// Wavefront-reduction / elementwise kernels of the EDRL head (fp32), gfx950.
//
// Reference semantics followed (file:line in /root/reference):
//   l2norm_axis1        F.normalize(z, dim=1) / F.normalize(z_proxy)        fusion_net.py:149-150
//   affine_bcast        mu_proxy + sigma_proxy * eps ; mu + U*sigma         fusion_net.py:143-146,907,910
//   topk_margin         mask / masked_select / topk(100) / exp margin       fusion_net.py:227-243
//   poe                 PoE.forward                                         fusion_net.py:26-52
//   kl_normal           KL_between_normals + get_KL_loss                    fusion_net.py:390-402,838-850
//   mha_core            nn.MultiheadAttention(1024, 8) inner attention      fusion_net.py:555,571
//   layernorm           nn.LayerNorm(1024)                                  fusion_net.py:560,573
//   bt_loss             DILR.bt_loss_cross + off_diagonal                   fusion_net.py:544-548,656-677
//   smooth_ce           label-smoothed cross entropy                        fusion_net.py:931-939
// All HBM-bound or latency-bound; coalesced along the innermost (feature) axis.
#include "edrl_common.h"

static inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

// ------------------------------------------------------------------ elementwise family
enum {
  EW_RELU = 0,        // out = max(a,0)
  EW_RELU_BWD = 1,    // out = a * (b > 0)
  EW_AXPBY = 2,       // out = alpha*a + beta*b
  EW_MUL = 3,         // out = a*b*alpha
  EW_SCALE = 4,       // out = alpha*a
  EW_SOFTPLUS = 5,    // out = softplus(a)   (torch: beta=1, threshold=20)
  EW_SOFTPLUS_BWD = 6,// out = a * sigmoid(b) (b = forward input), a where b>20
  EW_MASKED_BWD = 7,  // out = a * b * (c > 0)   (dy * dropout_mask * relu-mask of the output)
  EW_ADD_RELU = 8,    // out = max(a+b,0)
  EW_SCALE_BY_PTR = 9,// out = a * (*s) * alpha   (s = device scalar in b)
  EW_FILL = 10,       // out = alpha
  EW_LERP_BY_PTR = 11 // out = s*a + (1-s)*b, s = c[0] (device scalar)   (fusion_net.py:173)
};

__global__ __launch_bounds__(256) void ew_kernel(int op, long n, const float* __restrict__ a,
                                                 const float* __restrict__ b, const float* __restrict__ c,
                                                 float* __restrict__ out, float alpha, float beta) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = 0.f;
    switch (op) {
      case EW_RELU: v = fmaxf(a[i], 0.f); break;
      case EW_RELU_BWD: v = b[i] > 0.f ? a[i] : 0.f; break;
      case EW_AXPBY: v = alpha * a[i] + beta * b[i]; break;
      case EW_MUL: v = a[i] * b[i] * alpha; break;
      case EW_SCALE: v = alpha * a[i]; break;
      case EW_SOFTPLUS: { const float x = a[i]; v = x > 20.f ? x : log1pf(expf(x)); } break;
      case EW_SOFTPLUS_BWD: { const float x = b[i]; v = x > 20.f ? a[i] : a[i] / (1.f + expf(-x)); } break;
      case EW_MASKED_BWD: v = c[i] > 0.f ? a[i] * (b ? b[i] : 1.f) : 0.f; break;
      case EW_ADD_RELU: v = fmaxf(a[i] + b[i], 0.f); break;
      case EW_SCALE_BY_PTR: v = a[i] * b[0] * alpha; break;
      case EW_FILL: v = alpha; break;
      case EW_LERP_BY_PTR: v = c[0] * a[i] + (1.f - c[0]) * b[i]; break;
    }
    out[i] = v;
  }
}

// out = sum_i w[i] * (*in[i])   (loss mixers, fusion_net.py:870-879)
struct ScalarMixArgs { const float* in[8]; float w[8]; int n; };
__global__ void scalar_mix_kernel(ScalarMixArgs a, float* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    float s = 0.f;
    for (int i = 0; i < a.n; ++i) s += a.w[i] * a.in[i][0];
    out[0] = s;
  }
}

// ------------------------------------------------------------------ l2 normalise along axis 1 of [A][L][D]
__global__ __launch_bounds__(256) void l2norm_axis1_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               float* __restrict__ inv, long A, int L, int D,
                                                               float eps) {
  const long total = A * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long a = i / D;
    const int d = (int)(i - a * D);
    const float* px = x + a * L * D + d;
    float s = 0.f;
    for (int l = 0; l < L; ++l) { const float v = px[(long)l * D]; s += v * v; }
    const float nrm = sqrtf(s);
    const float iv = 1.f / fmaxf(nrm, eps);
    inv[i] = nrm > eps ? iv : -iv;  // sign bit marks the clamped (constant-denominator) case
    float* py = y + a * L * D + d;
    for (int l = 0; l < L; ++l) py[(long)l * D] = px[(long)l * D] * iv;
  }
}
__global__ __launch_bounds__(256) void l2norm_axis1_bwd_kernel(const float* __restrict__ dy,
                                                               const float* __restrict__ y,
                                                               const float* __restrict__ inv, float* __restrict__ dx,
                                                               long A, int L, int D) {
  const long total = A * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long a = i / D;
    const int d = (int)(i - a * D);
    const long base = a * L * D + d;
    const float ivs = inv[i];
    const float iv = fabsf(ivs);
    float dot = 0.f;
    if (ivs > 0.f)
      for (int l = 0; l < L; ++l) dot += dy[base + (long)l * D] * y[base + (long)l * D];
    for (int l = 0; l < L; ++l) dx[base + (long)l * D] = iv * (dy[base + (long)l * D] - y[base + (long)l * D] * dot);
  }
}

// Few columns, long axis (the proxy tensors: A*D = 512 columns, L = all tokens of the batch): one thread per column walks L serially
// and 2 workgroups run for ~0.3 ms.  Cooperative forms: a workgroup owns 16 columns, its 16 row lanes split L (row lane r takes
// l = r, r+16, ...: fixed order), partial sums combined in lane order through LDS (deterministic), second sweep by the same lanes.
#define CO_DL 16
#define CO_RL 16
template <int NS>
__device__ __forceinline__ void co_reduce(float (&s)[NS], float* sh, int rl, int dl) {
#pragma unroll
  for (int q = 0; q < NS; ++q) sh[(q * CO_RL + rl) * CO_DL + dl] = s[q];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < NS; ++q) {
    float t = 0.f;
    for (int r = 0; r < CO_RL; ++r) t += sh[(q * CO_RL + r) * CO_DL + dl];
    s[q] = t;
  }
}
__global__ __launch_bounds__(256) void l2norm_axis1_fwd_co_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                  float* __restrict__ inv, long A, int L, int D, float eps) {
  __shared__ float sh[CO_RL * CO_DL];
  const int dl = threadIdx.x % CO_DL, rl = threadIdx.x / CO_DL;
  const long i = (long)blockIdx.x * CO_DL + dl;
  const bool ok = i < A * D;
  const long a = ok ? i / D : 0;
  const int d = ok ? (int)(i - a * D) : 0;
  const float* px = x + a * L * D + d;
  float s[1] = {0.f};
  if (ok)
    for (int l = rl; l < L; l += CO_RL) { const float v = px[(long)l * D]; s[0] += v * v; }
  co_reduce<1>(s, sh, rl, dl);
  if (!ok) return;
  const float nrm = sqrtf(s[0]);
  const float iv = 1.f / fmaxf(nrm, eps);
  if (rl == 0) inv[i] = nrm > eps ? iv : -iv;
  float* py = y + a * L * D + d;
  for (int l = rl; l < L; l += CO_RL) py[(long)l * D] = px[(long)l * D] * iv;
}
__global__ __launch_bounds__(256) void l2norm_axis1_bwd_co_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                                  const float* __restrict__ inv, float* __restrict__ dx, long A, int L,
                                                                  int D) {
  __shared__ float sh[CO_RL * CO_DL];
  const int dl = threadIdx.x % CO_DL, rl = threadIdx.x / CO_DL;
  const long i = (long)blockIdx.x * CO_DL + dl;
  const bool ok = i < A * D;
  const long a = ok ? i / D : 0;
  const int d = ok ? (int)(i - a * D) : 0;
  const long base = a * L * D + d;
  const float ivs = ok ? inv[i] : 0.f;
  const float iv = fabsf(ivs);
  float s[1] = {0.f};
  if (ok && ivs > 0.f)
    for (int l = rl; l < L; l += CO_RL) s[0] += dy[base + (long)l * D] * y[base + (long)l * D];
  co_reduce<1>(s, sh, rl, dl);
  if (!ok) return;
  for (int l = rl; l < L; l += CO_RL) dx[base + (long)l * D] = iv * (dy[base + (long)l * D] - y[base + (long)l * D] * s[0]);
}
__global__ __launch_bounds__(256) void affine_bcast_bwd_co_kernel(const float* __restrict__ dout, const float* __restrict__ w,
                                                                  float* __restrict__ du, float* __restrict__ dv, long A, int L, int D) {
  __shared__ float sh[2 * CO_RL * CO_DL];
  const int dl = threadIdx.x % CO_DL, rl = threadIdx.x / CO_DL;
  const long i = (long)blockIdx.x * CO_DL + dl;
  const bool ok = i < A * D;
  const long a = ok ? i / D : 0;
  const int d = ok ? (int)(i - a * D) : 0;
  const long base = a * L * D + d;
  float s[2] = {0.f, 0.f};
  if (ok)
    for (int l = rl; l < L; l += CO_RL) {
      const float g = dout[base + (long)l * D];
      s[0] += g;
      s[1] += g * w[base + (long)l * D];
    }
  co_reduce<2>(s, sh, rl, dl);
  if (ok && rl == 0) { du[i] = s[0]; dv[i] = s[1]; }
}
// cooperative form when the column count leaves most of the chip idle and the serial walk is long
static inline bool co_shape(long A, int L, int D) { return A * D <= 256L * 128 && L >= 64; }

// out[a][l][d] = u[a][d] + v[a][d] * w[a][l][d]
__global__ __launch_bounds__(256) void affine_bcast_fwd_kernel(const float* __restrict__ u, const float* __restrict__ v,
                                                               const float* __restrict__ w, float* __restrict__ out,
                                                               long A, int L, int D) {
  const long total = A * L * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const long a = i / ((long)L * D);
    out[i] = u[a * D + d] + v[a * D + d] * w[i];
  }
}
// du[a][d] = sum_l dout ; dv[a][d] = sum_l dout*w
__global__ __launch_bounds__(256) void affine_bcast_bwd_kernel(const float* __restrict__ dout,
                                                               const float* __restrict__ w, float* __restrict__ du,
                                                               float* __restrict__ dv, long A, int L, int D) {
  const long total = A * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long a = i / D;
    const int d = (int)(i - a * D);
    const long base = a * L * D + d;
    float s0 = 0.f, s1 = 0.f;
    for (int l = 0; l < L; ++l) {
      const float g = dout[base + (long)l * D];
      s0 += g;
      s1 += g * w[base + (long)l * D];
    }
    du[i] = s0;
    dv[i] = s1;
  }
}

// ------------------------------------------------------------------ top-k margin loss (EPRL)
// att [B][C][S]; one block per (b, side): side 0 = row of the true class (S values),
// side 1 = the other C-1 rows ((C-1)*S values, in class order as masked_select yields them).
// Bitonic sort (value desc, index asc) in LDS; mean of the first K; sel[b][c][s] = 1 for chosen.
#define TOPK_MAX 4096
__global__ __launch_bounds__(256) void topk_margin_select_kernel(const float* __restrict__ att,
                                                                 const long long* __restrict__ y,
                                                                 unsigned char* __restrict__ sel,
                                                                 float* __restrict__ means, int B, int C, int S,
                                                                 int K, int NP) {
  __shared__ float sv[TOPK_MAX];
  __shared__ int si[TOPK_MAX];
  __shared__ float red[4];
  const int b = blockIdx.x >> 1, side = blockIdx.x & 1;
  const long long yl = y[b];
  const int tid = threadIdx.x;
  // A label outside [0, C) (deferred label check, medfusion.check_labels: the KeyError is raised later on the host) must not
  // index anything: the row contributes mean 0 and selects nothing, `sel` stays as the caller zeroed it.  Block-uniform exit.
  if (yl < 0 || yl >= C) {
    if (tid == 0) means[b * 2 + side] = 0.f;
    return;
  }
  const int yb = (int)yl;
  const int n = side == 0 ? S : (C - 1) * S;
  for (int i = tid; i < NP; i += 256) {
    float v = -INFINITY;
    int gi = 0x7fffffff;
    if (i < n) {
      int c, s;
      if (side == 0) { c = yb; s = i; }
      else { c = i / S; s = i - c * S; if (c >= yb) c += 1; }
      gi = c * S + s;
      v = att[(long)b * C * S + gi];
    }
    sv[i] = v; si[i] = gi;
  }
  __syncthreads();
  for (int k = 2; k <= NP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < NP; i += 256) {
        const int p = i ^ j;
        if (p > i) {
          const float v0 = sv[i], v1 = sv[p];
          const int i0 = si[i], i1 = si[p];
          // "before" order: larger value first, then smaller index
          const bool first_ok = (v0 > v1) || (v0 == v1 && i0 < i1);
          const bool desc = (i & k) == 0;
          if (desc ? !first_ok : first_ok) { sv[i] = v1; sv[p] = v0; si[i] = i1; si[p] = i0; }
        }
      }
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int i = tid; i < K; i += 256) {
    s += sv[i];
    sel[(long)b * C * S + si[i]] = 1;
  }
  s = edrl_block_sum_256(s, red);
  if (tid == 0) means[b * 2 + side] = s / (float)K;
}
// loss = mean_b exp(neg_mean - pos_mean);  e[b] saved for backward
__global__ __launch_bounds__(256) void topk_margin_final_kernel(const float* __restrict__ means, float* __restrict__ e,
                                                                float* __restrict__ loss, int B) {
  __shared__ float red[4];
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float v = expf(-means[b * 2] + means[b * 2 + 1]);
    e[b] = v;
    s += v;
  }
  s = edrl_block_sum_256(s, red);
  if (threadIdx.x == 0) loss[0] = s / (float)B;
}
__global__ __launch_bounds__(256) void topk_margin_bwd_kernel(const float* __restrict__ dloss,
                                                              const float* __restrict__ e,
                                                              const unsigned char* __restrict__ sel,
                                                              const long long* __restrict__ y, float* __restrict__ datt,
                                                              int B, int C, int S, int K) {
  const long total = (long)B * C * S;
  const float g = dloss[0] / ((float)B * (float)K);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int b = (int)(i / ((long)C * S));
    const int c = (int)((i / S) % C);
    float v = 0.f;
    if (sel[i]) v = (c == (int)y[b] ? -g : g) * e[b];
    datt[i] = v;
  }
}

// flag[0] |= 1 if any label is outside [0, C)   (reference raises KeyError, fusion_net.py:101,227)
__global__ void check_labels_kernel(const long long* __restrict__ y, int B, int C, int* __restrict__ flag) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x)
    if (y[b] < 0 || y[b] >= C) atomicOr(flag, 1);
}

// ------------------------------------------------------------------ PoE (2 experts), elementwise over R
// alpha = softmax(phi); T_i = 1/(s_i + eps); out = (sum mu_i a_i T_i + 1) / (sum a_i T_i)
__global__ __launch_bounds__(256) void poe_fwd_kernel(const float* __restrict__ mu0, const float* __restrict__ s0,
                                                      const float* __restrict__ mu1, const float* __restrict__ s1,
                                                      const float* __restrict__ phi, float* __restrict__ out, long R,
                                                      float eps) {
  const float m = fmaxf(phi[0], phi[1]);
  const float e0 = expf(phi[0] - m), e1 = expf(phi[1] - m);
  const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (long)gridDim.x * blockDim.x) {
    const float T0 = 1.f / (s0[i] + eps), T1 = 1.f / (s1[i] + eps);
    float tsum = 0.f, msum = 0.f;
    tsum += a0 * T0; msum += mu0[i] * a0 * T0;
    tsum += a1 * T1; msum += mu1[i] * a1 * T1;
    out[i] = msum / tsum + 1.f / tsum;
  }
}
// grid-stride over R; per-block partial dalpha written to part[block][2]
__global__ __launch_bounds__(256) void poe_bwd_kernel(const float* __restrict__ g, const float* __restrict__ mu0,
                                                      const float* __restrict__ s0, const float* __restrict__ mu1,
                                                      const float* __restrict__ s1, const float* __restrict__ phi,
                                                      float* __restrict__ dmu0, float* __restrict__ ds0,
                                                      float* __restrict__ dmu1, float* __restrict__ ds1,
                                                      float* __restrict__ part, long R, float eps) {
  __shared__ float red[4];
  const float m = fmaxf(phi[0], phi[1]);
  const float e0 = expf(phi[0] - m), e1 = expf(phi[1] - m);
  const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
  float da0 = 0.f, da1 = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < R; i += (long)gridDim.x * blockDim.x) {
    const float T0 = 1.f / (s0[i] + eps), T1 = 1.f / (s1[i] + eps);
    const float D = a0 * T0 + a1 * T1;
    const float Nn = mu0[i] * a0 * T0 + mu1[i] * a1 * T1;
    const float gi = g[i];
    const float dN = gi / D, dD = -gi * (Nn + 1.f) / (D * D);
    dmu0[i] = dN * a0 * T0;
    dmu1[i] = dN * a1 * T1;
    const float dT0 = dN * mu0[i] * a0 + dD * a0, dT1 = dN * mu1[i] * a1 + dD * a1;
    ds0[i] = -dT0 * T0 * T0;
    ds1[i] = -dT1 * T1 * T1;
    da0 += dN * mu0[i] * T0 + dD * T0;
    da1 += dN * mu1[i] * T1 + dD * T1;
  }
  da0 = edrl_block_sum_256(da0, red);
  da1 = edrl_block_sum_256(da1, red);
  if (threadIdx.x == 0) { part[blockIdx.x * 2] = da0; part[blockIdx.x * 2 + 1] = da1; }
}
__global__ void poe_bwd_final_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ phi,
                                     float* __restrict__ dphi) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double d0 = 0.0, d1 = 0.0;
    for (int i = 0; i < nblk; ++i) { d0 += part[i * 2]; d1 += part[i * 2 + 1]; }
    const float m = fmaxf(phi[0], phi[1]);
    const float e0 = expf(phi[0] - m), e1 = expf(phi[1] - m);
    const float a0 = e0 / (e0 + e1), a1 = e1 / (e0 + e1);
    const float dot = a0 * (float)d0 + a1 * (float)d1;
    dphi[0] = a0 * ((float)d0 - dot);
    dphi[1] = a1 * ((float)d1 - dot);
  }
}

// ------------------------------------------------------------------ KL(N(mu,sigma) || N(0,1)), summed over axis 1 of [B][C][D]
// loss = mean_{b,d} 0.5*( sum_c sigma^2 + sum_c mu^2 - C - sum_c 2 log(max(sigma,1e-8)) )
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sg,
                                                     float* __restrict__ part, long Bn, int C, int D) {
  __shared__ float red[4];
  const long total = Bn * D;
  float s = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long b = i / D;
    const int d = (int)(i - b * D);
    float fs = 0.f, ld = 0.f;
    for (int c = 0; c < C; ++c) {
      const float m = mu[(b * C + c) * D + d], sd = sg[(b * C + c) * D + d];
      fs += sd * sd + m * m;
      ld += 2.f * logf(fmaxf(sd, 1e-8f));
    }
    s += 0.5f * (fs - (float)C - ld);
  }
  s = edrl_block_sum_256(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void sum_partials_kernel(const float* __restrict__ part, int n, float scale, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += part[i];
    out[0] = (float)(s * (double)scale);
  }
}
__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ dloss, const float* __restrict__ mu,
                                                     const float* __restrict__ sg, float* __restrict__ dmu,
                                                     float* __restrict__ dsg, long n, float inv_count) {
  const float g = dloss[0] * inv_count;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float sd = sg[i];
    dmu[i] = g * mu[i];
    dsg[i] = g * (sd - (sd >= 1e-8f ? 1.f / sd : 0.f));
  }
}

// ------------------------------------------------------------------ multi-head attention core (short query)
// q [B][Lq][E], kv [B][N][2E] (keys | values), H heads of dh = E/H = 128.  One block per (b, h).
// P [B][H][Lq][N] saved for backward.  ctx [B][Lq][E].
#define MHA_DH 128
__global__ __launch_bounds__(256) void mha_core_fwd_kernel(const float* __restrict__ q, const float* __restrict__ kv,
                                                           float* __restrict__ P, float* __restrict__ ctx, int B,
                                                           int Lq, int N, int H, float scale) {
  extern __shared__ float sc[];  // [Lq][N]
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int E = H * MHA_DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* kb = kv + (long)b * N * 2 * E + h * MHA_DH;
  const float* vb = kb + E;
  for (int lq = 0; lq < Lq; ++lq) {
    const float* qp = q + ((long)b * Lq + lq) * E + h * MHA_DH;
    const float q0 = qp[lane], q1 = qp[lane + 64];
    for (int n = wave; n < N; n += 4) {
      const float* kp = kb + (long)n * 2 * E;
      float s = q0 * kp[lane] + q1 * kp[lane + 64];
      s = edrl_wave_sum(s);
      if (lane == 0) sc[lq * N + n] = s * scale;
    }
  }
  __syncthreads();
  for (int lq = wave; lq < Lq; lq += 4) {
    float m = -INFINITY;
    for (int n = lane; n < N; n += 64) m = fmaxf(m, sc[lq * N + n]);
    m = edrl_wave_max(m);
    float s = 0.f;
    for (int n = lane; n < N; n += 64) { const float e = expf(sc[lq * N + n] - m); sc[lq * N + n] = e; s += e; }
    s = edrl_wave_sum(s);
    const float inv = 1.f / s;
    float* pp = P + (((long)b * H + h) * Lq + lq) * N;
    for (int n = lane; n < N; n += 64) { const float p = sc[lq * N + n] * inv; sc[lq * N + n] = p; pp[n] = p; }
  }
  __syncthreads();
  for (int o = tid; o < Lq * MHA_DH; o += 256) {
    const int lq = o / MHA_DH, d = o - lq * MHA_DH;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += sc[lq * N + n] * vb[(long)n * 2 * E + d];
    ctx[((long)b * Lq + lq) * E + h * MHA_DH + d] = s;
  }
}
__global__ __launch_bounds__(256) void mha_core_bwd_kernel(const float* __restrict__ dctx, const float* __restrict__ q,
                                                           const float* __restrict__ kv, const float* __restrict__ P,
                                                           float* __restrict__ dq, float* __restrict__ dkv, int B,
                                                           int Lq, int N, int H, float scale) {
  extern __shared__ float sm[];  // dS [Lq][N], then Pn [Lq][N]
  float* dS = sm;
  float* Pn = sm + Lq * N;
  const int b = blockIdx.x / H, h = blockIdx.x - b * H;
  const int E = H * MHA_DH;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* kb = kv + (long)b * N * 2 * E + h * MHA_DH;
  const float* vb = kb + E;
  // dP[lq][n] = sum_d dctx[lq][d] * v[n][d]
  for (int lq = 0; lq < Lq; ++lq) {
    const float* gp = dctx + ((long)b * Lq + lq) * E + h * MHA_DH;
    const float g0 = gp[lane], g1 = gp[lane + 64];
    const float* pp = P + (((long)b * H + h) * Lq + lq) * N;
    for (int n = wave; n < N; n += 4) {
      const float* vp = vb + (long)n * 2 * E;
      float s = g0 * vp[lane] + g1 * vp[lane + 64];
      s = edrl_wave_sum(s);
      if (lane == 0) { dS[lq * N + n] = s; Pn[lq * N + n] = pp[n]; }
    }
  }
  __syncthreads();
  for (int lq = wave; lq < Lq; lq += 4) {
    float dot = 0.f;
    for (int n = lane; n < N; n += 64) dot += dS[lq * N + n] * Pn[lq * N + n];
    dot = edrl_wave_sum(dot);
    for (int n = lane; n < N; n += 64) dS[lq * N + n] = Pn[lq * N + n] * (dS[lq * N + n] - dot) * scale;
  }
  __syncthreads();
  // dq[lq][d] = sum_n dS[lq][n] * k[n][d]
  for (int o = tid; o < Lq * MHA_DH; o += 256) {
    const int lq = o / MHA_DH, d = o - lq * MHA_DH;
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += dS[lq * N + n] * kb[(long)n * 2 * E + d];
    dq[((long)b * Lq + lq) * E + h * MHA_DH + d] = s;
  }
  // dk[n][d] = sum_lq dS[lq][n] q[lq][d] ; dv[n][d] = sum_lq P[lq][n] dctx[lq][d]
  {
    const int d = tid & 127, isv = tid >> 7;
    float* ob = dkv + (long)b * N * 2 * E + h * MHA_DH + (isv ? E : 0);
    const float* src = isv ? dctx : q;
    const float* wt = isv ? Pn : dS;
    for (int n = 0; n < N; ++n) {
      float s = 0.f;
      for (int lq = 0; lq < Lq; ++lq) s += wt[lq * N + n] * src[((long)b * Lq + lq) * E + h * MHA_DH + d];
      ob[(long)n * 2 * E + d] = s;
    }
  }
}

// ------------------------------------------------------------------ LayerNorm over the last axis, one block per row
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bvec, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, int E,
                                                            float eps) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  const float* px = x + r * E;
  float s = 0.f;
  for (int i = threadIdx.x; i < E; i += 256) s += px[i];
  const float mu = edrl_block_sum_256(s, red) / (float)E;
  float v = 0.f;
  for (int i = threadIdx.x; i < E; i += 256) { const float d = px[i] - mu; v += d * d; }
  const float var = edrl_block_sum_256(v, red) / (float)E;
  const float rs = rsqrtf(var + eps);
  if (threadIdx.x == 0) { mean[r] = mu; rstd[r] = rs; }
  for (int i = threadIdx.x; i < E; i += 256) y[r * E + i] = (px[i] - mu) * rs * w[i] + bvec[i];
}
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ w, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            int E) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  const float mu = mean[r], rs = rstd[r];
  float s0 = 0.f, s1 = 0.f;
  for (int i = threadIdx.x; i < E; i += 256) {
    const float g = dy[r * E + i] * w[i];
    s0 += g;
    s1 += g * (x[r * E + i] - mu) * rs;
  }
  s0 = edrl_block_sum_256(s0, red) / (float)E;
  s1 = edrl_block_sum_256(s1, red) / (float)E;
  for (int i = threadIdx.x; i < E; i += 256) {
    const float g = dy[r * E + i] * w[i];
    const float xh = (x[r * E + i] - mu) * rs;
    dx[r * E + i] = rs * (g - s0 - xh * s1);
  }
}
// dw[i] = sum_r dy*xhat ; db[i] = sum_r dy
__global__ __launch_bounds__(256) void layernorm_bwd_params_kernel(const float* __restrict__ dy,
                                                                   const float* __restrict__ x,
                                                                   const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd,
                                                                   float* __restrict__ dw, float* __restrict__ db,
                                                                   long R, int E) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= E) return;
  float s0 = 0.f, s1 = 0.f;
  for (long r = 0; r < R; ++r) {
    const float g = dy[r * E + i];
    s0 += g;
    s1 += g * (x[r * E + i] - mean[r]) * rstd[r];
  }
  db[i] = s0;
  dw[i] = s1;
}

// ------------------------------------------------------------------ Barlow-twins style cross-correlation loss
// cc, cu: the [n][n] diagonal blocks of c (already divided by 4*batch).
// on_c = sum (c_ii-1)^2, off_c = sum_{i!=j} c_ij^2, on_u = sum c_ii^2, off_u likewise.
#define BT_BLOCKS 128
__global__ __launch_bounds__(256) void bt_loss_partial_kernel(const float* __restrict__ cc, const float* __restrict__ cu,
                                                              int n, float* __restrict__ part) {
  __shared__ float red[4];
  const long total = (long)n * n;
  float onc = 0.f, offc = 0.f, onu = 0.f, offu = 0.f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i - (long)r * n);
    const float a = cc[i], u = cu[i];
    if (r == c) { onc += (a - 1.f) * (a - 1.f); onu += u * u; }
    else { offc += a * a; offu += u * u; }
  }
  onc = edrl_block_sum_256(onc, red);
  offc = edrl_block_sum_256(offc, red);
  onu = edrl_block_sum_256(onu, red);
  offu = edrl_block_sum_256(offu, red);
  if (threadIdx.x == 0) {
    part[blockIdx.x * 4 + 0] = onc; part[blockIdx.x * 4 + 1] = offc;
    part[blockIdx.x * 4 + 2] = onu; part[blockIdx.x * 4 + 3] = offu;
  }
}
// out[0]=loss_c out[1]=on_c out[2]=off_c out[3]=loss_u out[4]=on_u out[5]=off_u out[6]=(loss_c+loss_u)/2
__global__ void bt_loss_final_kernel(const float* __restrict__ part, int nblk, float lambd, float* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    double s[4] = {0, 0, 0, 0};
    for (int i = 0; i < nblk; ++i)
      for (int k = 0; k < 4; ++k) s[k] += part[i * 4 + k];
    const float onc = (float)s[0], offc = (float)s[1], onu = (float)s[2], offu = (float)s[3];
    const float lc = onc + lambd * offc, lu = onu + lambd * offu;
    out[0] = lc; out[1] = onc; out[2] = offc; out[3] = lu; out[4] = onu; out[5] = offu;
    out[6] = (lc + lu) / 2.f;
  }
}
// d(loss12)/dc : dcc = g/2 * (2(c-1) diag | 2 lambda c off), dcu = g/2 * (2c diag | 2 lambda c off)
__global__ __launch_bounds__(256) void bt_loss_bwd_kernel(const float* __restrict__ dloss, const float* __restrict__ cc,
                                                          const float* __restrict__ cu, float* __restrict__ dcc,
                                                          float* __restrict__ dcu, int n, float lambd) {
  const long total = (long)n * n;
  const float g = dloss[0] * 0.5f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int r = (int)(i / n), c = (int)(i - (long)r * n);
    if (r == c) { dcc[i] = g * 2.f * (cc[i] - 1.f); dcu[i] = g * 2.f * cu[i]; }
    else { dcc[i] = g * 2.f * lambd * cc[i]; dcu[i] = g * 2.f * lambd * cu[i]; }
  }
}

// ------------------------------------------------------------------ label-smoothed cross entropy, pred [B][C]
__global__ __launch_bounds__(256) void smooth_ce_fwd_kernel(const float* __restrict__ pred,
                                                            const long long* __restrict__ y, float* __restrict__ loss,
                                                            int B, int C, float smoothing) {
  __shared__ float red[4];
  const float off = smoothing / (float)(C - 1), on = 1.f - smoothing;
  float s = 0.f;
  for (int b = threadIdx.x; b < B; b += 256) {
    const float* p = pred + (long)b * C;
    float m = -INFINITY;
    for (int k = 0; k < C; ++k) m = fmaxf(m, p[k]);
    float z = 0.f;
    for (int k = 0; k < C; ++k) z += expf(p[k] - m);
    const float lz = logf(z) + m;
    float acc = 0.f;
    for (int k = 0; k < C; ++k) acc += -(k == (int)y[b] ? on : off) * (p[k] - lz);
    s += acc;
  }
  s = edrl_block_sum_256(s, red);
  if (threadIdx.x == 0) loss[0] = s / (float)B;
}
__global__ __launch_bounds__(256) void smooth_ce_bwd_kernel(const float* __restrict__ dloss,
                                                            const float* __restrict__ pred,
                                                            const long long* __restrict__ y, float* __restrict__ dpred,
                                                            int B, int C, float smoothing) {
  const float off = smoothing / (float)(C - 1), on = 1.f - smoothing;
  const float tsum = off * (float)(C - 1) + on;
  const float g = dloss[0] / (float)B;
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    const float* p = pred + (long)b * C;
    float m = -INFINITY;
    for (int k = 0; k < C; ++k) m = fmaxf(m, p[k]);
    float z = 0.f;
    for (int k = 0; k < C; ++k) z += expf(p[k] - m);
    for (int k = 0; k < C; ++k) {
      const float sm = expf(p[k] - m) / z;
      dpred[(long)b * C + k] = g * (sm * tsum - (k == (int)y[b] ? on : off));
    }
  }
}

// argmax over the last axis (first maximum), int64 out — the `pred.argmax(dim=-1)` of fusion_train.py:213
__global__ void argmax_rows_kernel(const float* __restrict__ x, long long* __restrict__ out, int B, int C) {
  for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) {
    float best = x[(long)b * C];
    int bi = 0;
    for (int k = 1; k < C; ++k) { const float v = x[(long)b * C + k]; if (v > best) { best = v; bi = k; } }
    out[b] = bi;
  }
}


// ------------------------------------------------------------------ twin-view noise, KL/JS of soft labels
// out = clip(x + sigma * noise, 0, 1)   — the Gaussian high-noise view of data_harvard.py:769-783 on the device
// Salt-and-pepper noise (data_harvard.py:24-48): x[img, :, rows[img][j], cols[img][j]] = value for j < n_pts.
// The coordinates are INPUTS (drawn by the caller's RNG, as the reference draws them with numpy), so the result is
// bit-exact; duplicates write the same value, salt (1) is launched before pepper (0) as in the reference.
__global__ __launch_bounds__(256) void scatter_fill_nchw_kernel(float* __restrict__ x, const int* __restrict__ rows,
                                                                const int* __restrict__ cols, int n_img, long n_pts,
                                                                int C, int H, int W, float value) {
  const long total = (long)n_img * n_pts;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int img = (int)(i / n_pts);
    const int r = rows[i], c = cols[i];
    if ((unsigned)r >= (unsigned)H || (unsigned)c >= (unsigned)W) continue;
    float* base = x + ((long)img * C * H + r) * W + c;
    for (int ch = 0; ch < C; ++ch) base[(long)ch * H * W] = value;
  }
}

__global__ __launch_bounds__(256) void twin_view_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                                        float* __restrict__ out, long n, float sigma) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = fminf(fmaxf(x[i] + sigma * noise[i], 0.f), 1.f);
}
// compute_kl_divergence(p, m) = mean_b sum_c p log(p/m)   (code/MMD.py:92-95); single block
__global__ __launch_bounds__(256) void kl_rows_fwd_kernel(const float* __restrict__ p, const float* __restrict__ m,
                                                          float* __restrict__ out, int B, int C) {
  __shared__ float red[4];
  float s = 0.f;
  for (int i = threadIdx.x; i < B * C; i += 256) s += p[i] * logf(p[i] / m[i]);
  s = edrl_block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s / (float)B;
}
__global__ __launch_bounds__(256) void kl_rows_bwd_kernel(const float* __restrict__ dloss, const float* __restrict__ p,
                                                          const float* __restrict__ m, float* __restrict__ dp,
                                                          float* __restrict__ dm, int B, int C) {
  const float g = dloss[0] / (float)B;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < B * C; i += gridDim.x * blockDim.x) {
    dp[i] = g * (logf(p[i] / m[i]) + 1.f);
    dm[i] = -g * p[i] / m[i];
  }
}

// ------------------------------------------------------------------ eval-branch helpers (fusion_net.py:152-218)
// y[r][:] = softmax(x[r][:]); one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int C) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  const float* p = x + (long)row * C;
  float m = -INFINITY;
  for (int i = lane; i < C; i += 64) m = fmaxf(m, p[i]);
  m = edrl_wave_max(m);
  float s = 0.f;
  for (int i = lane; i < C; i += 64) s += expf(p[i] - m);
  s = edrl_wave_sum(s);
  for (int i = lane; i < C; i += 64) y[(long)row * C + i] = expf(p[i] - m) / s;
}
// out[r] = scale * sum_d x[r][d]; one wave per row
__global__ __launch_bounds__(256) void rowsum_kernel(const float* __restrict__ x, float* __restrict__ out, long R, int D,
                                                     long ld, float scale) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= R) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += x[row * ld + i];
  s = edrl_wave_sum(s);
  if (lane == 0) out[row] = s * scale;
}
// Pseudo-label selection (fusion_net.py:177-184): confidence/label = max/argmax over classes (first maximum),
// keep = confidence > threshold, and if nothing is kept the most confident sample is. count[0] = #kept.
__global__ void pseudo_label_kernel(const float* __restrict__ comb, int B, int C, float thr, long long* __restrict__ labels,
                                    unsigned char* __restrict__ keep, int* __restrict__ count) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int n = 0, best_b = 0;
  float best_c = -INFINITY;
  for (int b = 0; b < B; ++b) {
    float cf = comb[(long)b * C];
    int lb = 0;
    for (int k = 1; k < C; ++k) { const float v = comb[(long)b * C + k]; if (v > cf) { cf = v; lb = k; } }
    labels[b] = lb;
    keep[b] = cf > thr ? 1 : 0;
    n += keep[b];
    if (cf > best_c) { best_c = cf; best_b = b; }
  }
  if (n == 0) { keep[best_b] = 1; n = 1; }
  count[0] = n;
}
// mean_r ( - sum_k softmax(x_r)_k * log_softmax(x_r)_k )    (EPRL.entropy_regularization, fusion_net.py:127-131)
__global__ __launch_bounds__(256) void entropy_rows_kernel(const float* __restrict__ x, float* __restrict__ out, int R, int C) {
  __shared__ float red[4];
  float s = 0.f;
  for (int r = threadIdx.x; r < R; r += 256) {
    const float* p = x + (long)r * C;
    float m = -INFINITY;
    for (int k = 0; k < C; ++k) m = fmaxf(m, p[k]);
    float z = 0.f;
    for (int k = 0; k < C; ++k) z += expf(p[k] - m);
    const float lz = logf(z) + m;
    float e = 0.f;
    for (int k = 0; k < C; ++k) e -= expf(p[k] - lz) * (p[k] - lz);
    s += e;
  }
  s = edrl_block_sum_256(s, red);
  if (threadIdx.x == 0) out[0] = s / (float)R;
}
// eval-mode BatchNorm as the (mean, scale, shift) triple of edrl_bn_apply_f32: scale = gamma / sqrt(var + eps)
__global__ void bn_eval_params_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rv, float eps, float* __restrict__ scale,
                                      float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  scale[c] = (gamma ? gamma[c] : 1.f) / sqrtf(rv[c] + eps);
  shift[c] = beta ? beta[c] : 0.f;
}

extern "C" {

int edrl_scatter_fill_nchw_f32(float* x, const int* rows, const int* cols, int n_img, long n_pts, int C, int H, int W,
                               float value, hipStream_t st) {
  if (n_img <= 0 || n_pts < 0 || C <= 0 || H <= 0 || W <= 0) return EDRL_EINVAL;
  if (n_pts == 0) return 0;
  hipLaunchKernelGGL(scatter_fill_nchw_kernel, dim3(ew_grid((long)n_img * n_pts)), dim3(256), 0, st, x, rows, cols, n_img, n_pts,
                     C, H, W, value);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_twin_view_f32(const float* x, const float* noise, float* out, long n, float sigma, hipStream_t st) {
  if (n <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(twin_view_kernel, dim3(ew_grid(n)), dim3(256), 0, st, x, noise, out, n, sigma);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_kl_rows_fwd_f32(const float* p, const float* m, float* out, int B, int C, hipStream_t st) {
  if (B <= 0 || C <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(kl_rows_fwd_kernel, dim3(1), dim3(256), 0, st, p, m, out, B, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_kl_rows_bwd_f32(const float* dloss, const float* p, const float* m, float* dp, float* dm, int B, int C,
                         hipStream_t st) {
  if (B <= 0 || C <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(kl_rows_bwd_kernel, dim3(ew_grid((long)B * C)), dim3(256), 0, st, dloss, p, m, dp, dm, B, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_softmax_rows_f32(const float* x, float* y, int R, int C, hipStream_t st) {
  if (R <= 0 || C <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(edrl_cdiv(R, 4)), dim3(256), 0, st, x, y, R, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_rowsum_f32(const float* x, float* out, long R, int D, long ld, float scale, hipStream_t st) {
  if (R <= 0 || D <= 0 || ld < D) return EDRL_EINVAL;
  hipLaunchKernelGGL(rowsum_kernel, dim3(edrl_cdiv(R, 4)), dim3(256), 0, st, x, out, R, D, ld, scale);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_pseudo_label_f32(const float* comb, int B, int C, float threshold, long long* labels, unsigned char* keep,
                          int* count, hipStream_t st) {
  if (B <= 0 || C <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(pseudo_label_kernel, dim3(1), dim3(64), 0, st, comb, B, C, threshold, labels, keep, count);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_entropy_rows_f32(const float* x, float* out, int R, int C, hipStream_t st) {
  if (R <= 0 || C <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(entropy_rows_kernel, dim3(1), dim3(256), 0, st, x, out, R, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_bn_eval_params_f32(const float* gamma, const float* beta, const float* running_var, float eps, float* scale,
                            float* shift, int C, hipStream_t st) {
  if (C <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(bn_eval_params_kernel, dim3(edrl_cdiv(C, 256)), dim3(256), 0, st, gamma, beta, running_var, eps,
                     scale, shift, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}


int edrl_ew_f32(int op, long n, const float* a, const float* b, const float* c, float* out, float alpha,
                float beta, hipStream_t st) {
  if (n < 0 || op < 0 || op > EW_LERP_BY_PTR) return EDRL_EINVAL;
  if (n == 0) return 0;
  hipLaunchKernelGGL(ew_kernel, dim3(ew_grid(n)), dim3(256), 0, st, op, n, a, b, c, out, alpha, beta);
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_scalar_mix_f32(const float* const* in, const float* w, int n, float* out, hipStream_t st) {
  if (n <= 0 || n > 8) return EDRL_EINVAL;
  ScalarMixArgs a;
  a.n = n;
  for (int i = 0; i < n; ++i) { a.in[i] = in[i]; a.w[i] = w[i]; }
  for (int i = n; i < 8; ++i) { a.in[i] = nullptr; a.w[i] = 0.f; }
  hipLaunchKernelGGL(scalar_mix_kernel, dim3(1), dim3(64), 0, st, a, out);
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_l2norm_axis1_fwd_f32(const float* x, float* y, float* inv, long A, int L, int D, float eps,
                              hipStream_t st) {
  if (A <= 0 || L <= 0 || D <= 0) return EDRL_EINVAL;
  if (co_shape(A, L, D))
    hipLaunchKernelGGL(l2norm_axis1_fwd_co_kernel, dim3((unsigned)edrl_cdiv(A * D, CO_DL)), dim3(256), 0, st, x, y, inv, A, L, D, eps);
  else
    hipLaunchKernelGGL(l2norm_axis1_fwd_kernel, dim3(ew_grid(A * D)), dim3(256), 0, st, x, y, inv, A, L, D, eps);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_l2norm_axis1_bwd_f32(const float* dy, const float* y, const float* inv, float* dx, long A, int L, int D,
                              hipStream_t st) {
  if (A <= 0 || L <= 0 || D <= 0) return EDRL_EINVAL;
  if (co_shape(A, L, D))
    hipLaunchKernelGGL(l2norm_axis1_bwd_co_kernel, dim3((unsigned)edrl_cdiv(A * D, CO_DL)), dim3(256), 0, st, dy, y, inv, dx, A, L, D);
  else
    hipLaunchKernelGGL(l2norm_axis1_bwd_kernel, dim3(ew_grid(A * D)), dim3(256), 0, st, dy, y, inv, dx, A, L, D);
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_affine_bcast_fwd_f32(const float* u, const float* v, const float* w, float* out, long A, int L, int D,
                              hipStream_t st) {
  if (A <= 0 || L <= 0 || D <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(affine_bcast_fwd_kernel, dim3(ew_grid(A * L * D)), dim3(256), 0, st, u, v, w, out, A, L, D);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_affine_bcast_bwd_f32(const float* dout, const float* w, float* du, float* dv, long A, int L, int D,
                              hipStream_t st) {
  if (A <= 0 || L <= 0 || D <= 0) return EDRL_EINVAL;
  if (co_shape(A, L, D))
    hipLaunchKernelGGL(affine_bcast_bwd_co_kernel, dim3((unsigned)edrl_cdiv(A * D, CO_DL)), dim3(256), 0, st, dout, w, du, dv, A, L, D);
  else
    hipLaunchKernelGGL(affine_bcast_bwd_kernel, dim3(ew_grid(A * D)), dim3(256), 0, st, dout, w, du, dv, A, L, D);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// att [B][C][S], y int64 [B] in [0,C).  sel [B][C][S] must be zeroed by the caller.
// means [B][2], e [B], loss [1].
int edrl_topk_margin_fwd_f32(const float* att, const long long* y, unsigned char* sel, float* means, float* e,
                             float* loss, int B, int C, int S, int K, hipStream_t st) {
  if (B <= 0 || C < 2 || S <= 0 || K <= 0 || K > S || (long)(C - 1) * S > TOPK_MAX) return EDRL_EINVAL;
  int np = 1;
  while (np < (C - 1) * S || np < S) np <<= 1;
  hipLaunchKernelGGL(topk_margin_select_kernel, dim3(B * 2), dim3(256), 0, st, att, y, sel, means, B, C, S, K, np);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(topk_margin_final_kernel, dim3(1), dim3(256), 0, st, means, e, loss, B);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_topk_margin_bwd_f32(const float* dloss, const float* e, const unsigned char* sel, const long long* y,
                             float* datt, int B, int C, int S, int K, hipStream_t st) {
  if (B <= 0 || C < 2 || S <= 0 || K <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(topk_margin_bwd_kernel, dim3(ew_grid((long)B * C * S)), dim3(256), 0, st, dloss, e, sel, y,
                     datt, B, C, S, K);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_check_labels(const long long* y, int B, int C, int* flag, hipStream_t st) {
  if (B <= 0 || C <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(check_labels_kernel, dim3(1), dim3(256), 0, st, y, B, C, flag);
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_poe2_fwd_f32(const float* mu0, const float* s0, const float* mu1, const float* s1, const float* phi,
                      float* out, long R, float eps, hipStream_t st) {
  if (R <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(poe_fwd_kernel, dim3(ew_grid(R)), dim3(256), 0, st, mu0, s0, mu1, s1, phi, out, R, eps);
  EDRL_LAUNCH_CHECK();
  return 0;
}
// workspace: >= 2*64 floats
int edrl_poe2_bwd_f32(const float* g, const float* mu0, const float* s0, const float* mu1, const float* s1,
                      const float* phi, float* dmu0, float* ds0, float* dmu1, float* ds1, float* dphi,
                      float* workspace, long R, float eps, hipStream_t st) {
  if (R <= 0) return EDRL_EINVAL;
  int nblk = ew_grid(R);
  if (nblk > 64) nblk = 64;
  hipLaunchKernelGGL(poe_bwd_kernel, dim3(nblk), dim3(256), 0, st, g, mu0, s0, mu1, s1, phi, dmu0, ds0, dmu1, ds1,
                     workspace, R, eps);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(poe_bwd_final_kernel, dim3(1), dim3(64), 0, st, workspace, nblk, phi, dphi);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// mu, sg [Bn][C][D]; workspace >= 64 floats
int edrl_kl_normal_fwd_f32(const float* mu, const float* sg, float* loss, float* workspace, long Bn, int C, int D,
                           hipStream_t st) {
  if (Bn <= 0 || C <= 0 || D <= 0) return EDRL_EINVAL;
  int nblk = ew_grid(Bn * D);
  if (nblk > 64) nblk = 64;
  hipLaunchKernelGGL(kl_fwd_kernel, dim3(nblk), dim3(256), 0, st, mu, sg, workspace, Bn, C, D);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, workspace, nblk, 1.f / (float)(Bn * D), loss);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_kl_normal_bwd_f32(const float* dloss, const float* mu, const float* sg, float* dmu, float* dsg, long Bn,
                           int C, int D, hipStream_t st) {
  if (Bn <= 0 || C <= 0 || D <= 0) return EDRL_EINVAL;
  const long n = Bn * C * D;
  hipLaunchKernelGGL(kl_bwd_kernel, dim3(ew_grid(n)), dim3(256), 0, st, dloss, mu, sg, dmu, dsg, n,
                     1.f / (float)(Bn * D));
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_mha_core_fwd_f32(const float* q, const float* kv, float* P, float* ctx, int B, int Lq, int N, int H, int E,
                          hipStream_t st) {
  if (B <= 0 || Lq <= 0 || N <= 0 || H <= 0 || E != H * MHA_DH || (long)Lq * N * 4 > 60000) return EDRL_EINVAL;
  hipLaunchKernelGGL(mha_core_fwd_kernel, dim3(B * H), dim3(256), (size_t)Lq * N * sizeof(float), st, q, kv, P, ctx,
                     B, Lq, N, H, 1.f / sqrtf((float)MHA_DH));
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_mha_core_bwd_f32(const float* dctx, const float* q, const float* kv, const float* P, float* dq, float* dkv,
                          int B, int Lq, int N, int H, int E, hipStream_t st) {
  if (B <= 0 || Lq <= 0 || N <= 0 || H <= 0 || E != H * MHA_DH || (long)Lq * N * 8 > 60000) return EDRL_EINVAL;
  hipLaunchKernelGGL(mha_core_bwd_kernel, dim3(B * H), dim3(256), (size_t)2 * Lq * N * sizeof(float), st, dctx, q,
                     kv, P, dq, dkv, B, Lq, N, H, 1.f / sqrtf((float)MHA_DH));
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_layernorm_fwd_f32(const float* x, const float* w, const float* b, float* y, float* mean, float* rstd,
                           long R, int E, float eps, hipStream_t st) {
  if (R <= 0 || E <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3((unsigned)R), dim3(256), 0, st, x, w, b, y, mean, rstd, E, eps);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_layernorm_bwd_f32(const float* dy, const float* x, const float* w, const float* mean, const float* rstd,
                           float* dx, float* dw, float* db, long R, int E, hipStream_t st) {
  if (R <= 0 || E <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3((unsigned)R), dim3(256), 0, st, dy, x, w, mean, rstd, dx, E);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(layernorm_bwd_params_kernel, dim3(edrl_cdiv(E, 256)), dim3(256), 0, st, dy, x, mean, rstd, dw,
                     db, R, E);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// cc, cu: [n][n]; out: 7 floats (see kernel); workspace >= 4*BT_BLOCKS floats
int edrl_bt_loss_fwd_f32(const float* cc, const float* cu, int n, float lambd, float* out, float* workspace,
                         hipStream_t st) {
  if (n <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(bt_loss_partial_kernel, dim3(BT_BLOCKS), dim3(256), 0, st, cc, cu, n, workspace);
  EDRL_LAUNCH_CHECK();
  hipLaunchKernelGGL(bt_loss_final_kernel, dim3(1), dim3(64), 0, st, workspace, BT_BLOCKS, lambd, out);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_bt_loss_bwd_f32(const float* dloss12, const float* cc, const float* cu, float* dcc, float* dcu, int n,
                         float lambd, hipStream_t st) {
  if (n <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(bt_loss_bwd_kernel, dim3(ew_grid((long)n * n)), dim3(256), 0, st, dloss12, cc, cu, dcc, dcu, n,
                     lambd);
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_smooth_ce_fwd_f32(const float* pred, const long long* y, float* loss, int B, int C, float smoothing,
                           hipStream_t st) {
  if (B <= 0 || C < 2) return EDRL_EINVAL;
  hipLaunchKernelGGL(smooth_ce_fwd_kernel, dim3(1), dim3(256), 0, st, pred, y, loss, B, C, smoothing);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_smooth_ce_bwd_f32(const float* dloss, const float* pred, const long long* y, float* dpred, int B, int C,
                           float smoothing, hipStream_t st) {
  if (B <= 0 || C < 2) return EDRL_EINVAL;
  hipLaunchKernelGGL(smooth_ce_bwd_kernel, dim3(edrl_cdiv(B, 256)), dim3(256), 0, st, dloss, pred, y, dpred, B, C,
                     smoothing);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_argmax_rows_f32(const float* x, long long* out, int B, int C, hipStream_t st) {
  if (B <= 0 || C <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(argmax_rows_kernel, dim3(edrl_cdiv(B, 256)), dim3(256), 0, st, x, out, B, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
