// Does the fp32 MFMA (v_mfma_f32_32x32x2_f32) share the SIMD's execution lanes with the fp32 VALU on gfx950?
// Each wave runs ITER iterations of {NM independent MFMAs + V independent v_fma_f32}; one wave per SIMD (1024 waves) or two
// waves per SIMD where the partner wave runs ONLY VALU work.  If the two pipes were independent the time per MFMA would stay
// at 64 cycles until V*4 > 64; if they share lanes it grows as 64 + 4*V (same wave) and a VALU-only partner slows the MFMA wave.
// build: hipcc -O3 --offload-arch=gfx950 mfma_valu_f32.hip -o mfma_valu_f32     run: ./mfma_valu_f32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITER = 16384, NM = 4;

template <int V, int MODE>   // MODE 0: every wave MFMA + V VALU per MFMA; MODE 1: waves 0-3 MFMA only, waves 4-7 (their SIMD co-residents) VALU only, V per MFMA slot
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc) {
  const int wave = threadIdx.x >> 6;
  // waves are dealt to the 4 SIMDs round-robin: waves 0-3 sit one per SIMD, waves 4-7 are their co-residents
  const bool do_mfma = MODE == 0 || (wave >> 2) == 0;
  const bool do_valu = MODE == 0 || (wave >> 2) == 1;
  f32x16 acc[NM];
  for (int m = 0; m < NM; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = a + i;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if (do_mfma) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
      if (do_valu) {
#pragma unroll
        for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[v & 7]) : "v"(b));
      }
    }
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int m = 0; m < NM; ++m) s += acc[m][0];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}

template <int V, int MODE>
static void run(const char* label, int waves_per_simd) {
  const int threads = 256 * waves_per_simd, blocks = 256;      // one workgroup per CU, waves_per_simd waves on each SIMD
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * threads * blocks);
  hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<V, MODE><<<blocks, threads>>>(out, cyc);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<V, MODE><<<blocks, threads>>>(out, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), cyc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
  double cm = 0; int nm = 0;
  for (size_t i = 0; i < h.size(); ++i) if (MODE == 0 || (i % (threads / 64)) < 4) { cm += h[i]; ++nm; }   // the MFMA waves
  // s_memtime / readcyclecounter counts shader cycles here (64.0 per MFMA in the MFMA-only run = the documented issue interval)
  const double cyc_wave = cm / nm / ((double)ITER * NM), ns_wall = ms * 1e6 / ((double)ITER * NM);
  printf("%-64s V=%2d  MFMA wave: %6.1f cycles per MFMA   (kernel wall %6.2f ns per MFMA slot)\n", label, V, cyc_wave, ns_wall);
  hipFree(out); hipFree(cyc);
}

int main() {
  printf("v_mfma_f32_32x32x2_f32 issue interval: 64 cycles/SIMD (MI355X_MICROARCH.md); ITER=%d x %d MFMAs per wave\n", ITER, NM);
  run<0, 0>("one wave/SIMD, MFMA only", 1);
  run<2, 0>("one wave/SIMD, MFMA + V v_fma_f32 per MFMA (same wave)", 1);
  run<4, 0>("one wave/SIMD, MFMA + V v_fma_f32 per MFMA (same wave)", 1);
  run<8, 0>("one wave/SIMD, MFMA + V v_fma_f32 per MFMA (same wave)", 1);
  run<16, 0>("one wave/SIMD, MFMA + V v_fma_f32 per MFMA (same wave)", 1);
  run<0, 0>("two waves/SIMD, both MFMA only", 2);
  run<4, 1>("two waves/SIMD: MFMA-only wave beside a VALU-only wave (V/slot)", 2);
  run<8, 1>("two waves/SIMD: MFMA-only wave beside a VALU-only wave (V/slot)", 2);
  run<16, 1>("two waves/SIMD: MFMA-only wave beside a VALU-only wave (V/slot)", 2);
  return 0;
}
