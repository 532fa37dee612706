// Fused multi-tensor Adam: the optimiser step of fusion_train.py:224 / :747
// (torch.optim.Adam(lr, betas=(0.9,0.999), eps=1e-8, weight_decay=1e-6): L2 decay folded into the gradient, no amsgrad)
// over ALL parameter tensors in one launch.  HBM-bound: 16 B read + 12 B written per element, one pass (the stock
// foreach implementation makes ~7 passes).  Work decomposition: a table of tensors (pointers change every step because
// autograd re-allocates the gradients) and a table of fixed-size chunks (depends on the sizes only, cached by the host).
#include "edrl_common.h"
#include <math.h>
#include <stdint.h>

struct AdamTensor {
  float* p;
  const float* g;
  float* m;
  float* v;
  long n;
};
struct AdamChunk {
  int tensor;
  int chunk;
};
#define ADAM_CHUNK 16384   // elements per workgroup: 256 threads x 16 float4

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTensor* __restrict__ tab, const AdamChunk* __restrict__ chunks,
                                                         float lr_over_bc1, float w1, float beta2, float w2, float eps,
                                                         float wd, float bc2_sqrt) {
  const AdamChunk c = chunks[blockIdx.x];
  const AdamTensor t = tab[c.tensor];
  const long base = (long)c.chunk * ADAM_CHUNK;
  long end = base + ADAM_CHUNK; if (end > t.n) end = t.n;
  auto upd = [&](float& p, float g, float& m, float& v) {
    g = g + wd * p;                       // grad.add(param, alpha=weight_decay)
    m = m + w1 * (g - m);                 // exp_avg.lerp_(grad, 1-beta1)
    v = v * beta2 + w2 * g * g;           // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - lr_over_bc1 * (m / denom);    // param.addcdiv_(exp_avg, denom, value=-lr/bias_correction1)
  };
  const bool vec = ((((uintptr_t)t.p | (uintptr_t)t.g | (uintptr_t)t.m | (uintptr_t)t.v) & 15) == 0);
  if (vec) {
    const long end4 = base + ((end - base) & ~3L);
    for (long i = base + threadIdx.x * 4; i < end4; i += 256 * 4) {
      f32x4 p = *reinterpret_cast<const f32x4*>(t.p + i);
      const f32x4 g = *reinterpret_cast<const f32x4*>(t.g + i);
      f32x4 m = *reinterpret_cast<const f32x4*>(t.m + i);
      f32x4 v = *reinterpret_cast<const f32x4*>(t.v + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = p[e], me = m[e], ve = v[e];
        upd(pe, g[e], me, ve);
        p[e] = pe; m[e] = me; v[e] = ve;
      }
      *reinterpret_cast<f32x4*>(t.p + i) = p;
      *reinterpret_cast<f32x4*>(t.m + i) = m;
      *reinterpret_cast<f32x4*>(t.v + i) = v;
    }
    for (long i = end4 + threadIdx.x; i < end; i += 256) upd(t.p[i], t.g[i], t.m[i], t.v[i]);
  } else {
    for (long i = base + threadIdx.x; i < end; i += 256) upd(t.p[i], t.g[i], t.m[i], t.v[i]);
  }
}

// Weight shadows of a whole trunk in ONE launch (round 5; replaces one cast + one permute launch per conv layer, view and step):
// for every conv weight w fp32 [A][B][C] (= [Co][KH*KW][Ci]) the forward operand `cast` (same layout, bf16: bf16 trunk only) and
// the data-gradient operand `perm` [C][B][A] (= [Ci][KH*KW][Co]; bf16 or fp32).  Same table + chunk scheme as the Adam kernel;
// the values are exactly those of edrl_cast_f32_to_bf16 / edrl_permute_weight_{f32,bf16} (one rounding of the fp32 weight).
struct ShadowTensor {
  const float* w;
  void* cast;        // bf16 [A][B][C], or NULL
  void* perm;        // [C][B][A] bf16 (perm_bf16 != 0) or fp32, or NULL
  int A, B, C, perm_bf16;
  long n;
};
__global__ __launch_bounds__(256) void weight_shadow_multi_kernel(const ShadowTensor* __restrict__ tab, const AdamChunk* __restrict__ chunks) {
  const AdamChunk c = chunks[blockIdx.x];
  const ShadowTensor t = tab[c.tensor];
  const long base = (long)c.chunk * ADAM_CHUNK;
  long end = base + ADAM_CHUNK; if (end > t.n) end = t.n;
  __bf16* cast = (__bf16*)t.cast;
  const long BC = (long)t.B * t.C;
  for (long i = base + threadIdx.x; i < end; i += 256) {
    const float v = t.w[i];
    if (cast) cast[i] = (__bf16)v;
    if (t.perm) {
      const int a = (int)(i / BC);
      const long r = i - (long)a * BC;
      const int b = (int)(r / t.C), cc = (int)(r - (long)b * t.C);
      const long o = ((long)cc * t.B + b) * t.A + a;
      if (t.perm_bf16) ((__bf16*)t.perm)[o] = (__bf16)v; else ((float*)t.perm)[o] = v;
    }
  }
}

extern "C" {

int edrl_adam_chunk_elems(void) { return ADAM_CHUNK; }

// tensors: device array of n_tensors records {const float* w; void* cast; void* perm; int A, B, C, perm_bf16; long n} (48 bytes
// each, n = A*B*C); chunks: device array of {int tensor; int chunk} records covering every tensor in edrl_adam_chunk_elems() pieces.
int edrl_weight_shadows_multi(const void* tensors, int n_tensors, const void* chunks, int n_chunks, hipStream_t st) {
  if (n_tensors <= 0 || n_chunks <= 0 || tensors == nullptr || chunks == nullptr) return EDRL_EINVAL;
  hipLaunchKernelGGL(weight_shadow_multi_kernel, dim3(n_chunks), dim3(256), 0, st, (const ShadowTensor*)tensors, (const AdamChunk*)chunks);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// tensors: device array of n_tensors records {float* p; const float* g; float* m; float* v; long n} (40 bytes each);
// chunks: device array of n_chunks records {int tensor; int chunk} covering every tensor in ADAM_CHUNK-element pieces.
// step >= 1 is the step count AFTER this update (torch: state['step'] += 1 first); bias corrections are formed in double.
int edrl_adam_multi_f32(const void* tensors, int n_tensors, const void* chunks, int n_chunks, double lr, double beta1,
                        double beta2, double eps, double weight_decay, long step, hipStream_t st) {
  if (n_tensors <= 0 || n_chunks <= 0 || step < 1 || tensors == nullptr || chunks == nullptr) return EDRL_EINVAL;
  // hyper-parameters arrive as doubles and are rounded to fp32 where torch rounds them (1-beta as a double first)
  const double bc1 = 1.0 - pow(beta1, (double)step);
  const double bc2 = 1.0 - pow(beta2, (double)step);
  hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, st, (const AdamTensor*)tensors, (const AdamChunk*)chunks,
                     (float)(lr / bc1), (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps,
                     (float)weight_decay, (float)sqrt(bc2));
  EDRL_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
