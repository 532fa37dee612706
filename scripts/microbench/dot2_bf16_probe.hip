// Probe of v_dot2c_f32_bf16 on gfx950: r = (h0, h1) . (s0, s1) + c with constant selectors (inline / literal operands) and with
// the same selectors passed at run time.  build: hipcc -O3 --offload-arch=gfx950 dot2_bf16_probe.hip -o dot2_bf16_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__global__ void k(const float* a, float* o, unsigned rs0, unsigned rs1) {
  const int i = threadIdx.x;
  const float a0 = a[2 * i], a1 = a[2 * i + 1];
  bf16x2 h; h[0] = (__bf16)a0; h[1] = (__bf16)a1;
  const bf16x2 sel0 = {(__bf16)-1.f, (__bf16)0.f}, sel1 = {(__bf16)0.f, (__bf16)-1.f};
  o[8 * i + 0] = __builtin_amdgcn_fdot2_f32_bf16(h, sel0, a0, false);
  o[8 * i + 1] = __builtin_amdgcn_fdot2_f32_bf16(h, sel1, a1, false);
  o[8 * i + 2] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, rs0), a0, false);
  o[8 * i + 3] = __builtin_amdgcn_fdot2_f32_bf16(h, __builtin_bit_cast(bf16x2, rs1), a1, false);
  o[8 * i + 4] = a0 - (float)h[0];
  o[8 * i + 5] = a1 - (float)h[1];
  o[8 * i + 6] = (float)h[0];
  o[8 * i + 7] = (float)h[1];
}
int main() {
  const int n = 8;
  float ha[2 * n] = {1.2345678f, -7.654321f, 3.0e38f, -1e-30f, 100.125f, 0.33333334f, 65537.0f, -2.5f,
                     1.0f, 1.0039062f, 1e-20f, 123456.789f, -0.001f, 5e10f, 7.0f, 9.99f};
  float *da, *d;
  hipMalloc(&da, sizeof(ha)); hipMalloc(&d, 8 * n * sizeof(float));
  hipMemcpy(da, ha, sizeof(ha), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(n), 0, 0, da, d, 0x0000BF80u, 0xBF800000u);
  float ho[8 * n];
  hipMemcpy(ho, d, sizeof(ho), hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i)
    printf("a=(%.9g, %.9g) h=(%.9g, %.9g)  const: %.9g %.9g  runtime: %.9g %.9g  ref: %.9g %.9g\n", ha[2 * i], ha[2 * i + 1], ho[8 * i + 6],
           ho[8 * i + 7], ho[8 * i], ho[8 * i + 1], ho[8 * i + 2], ho[8 * i + 3], ho[8 * i + 4], ho[8 * i + 5]);
  return 0;
}
