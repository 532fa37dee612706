// Microtest (gfx950): what does an LDS-DMA buffer load (buffer_load_dwordx4 ... offen lds) write for a lane whose offset fails the
// descriptor's range check -- zeros, or nothing (LDS keeps its old bytes)?  The conv gather relies on the answer for padding taps.
// build: hipcc --offload-arch=gfx950 -O3 ldsdma_oob.hip -o ldsdma_oob ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
__global__ void k1(const char* src, int* dst, int nbytes, const int* offs) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  const int off = offs[threadIdx.x];
  for (int i = threadIdx.x; i < 4096 / 4; i += blockDim.x) ((int*)smem)[i] = 0x7f7f7f7f;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + wave * 1024), 16, off, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int i = threadIdx.x; i < 4096 / 4; i += blockDim.x) dst[i] = ((int*)smem)[i];
}
int main() {
  const int n = 4096;
  std::vector<int> h(n / 4);
  for (int i = 0; i < n / 4; ++i) h[i] = 0x1000 + i;
  std::vector<int> offs(256);
  for (int t = 0; t < 256; ++t) offs[t] = (t % 3 == 1) ? (int)0x80000000u : ((t % 3 == 2) ? n + 64 : t * 16);
  char* dsrc; int* ddst; int* doffs;
  hipMalloc(&dsrc, n); hipMalloc(&ddst, n); hipMalloc(&doffs, 1024);
  hipMemcpy(dsrc, h.data(), n, hipMemcpyHostToDevice);
  hipMemcpy(doffs, offs.data(), 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k1, dim3(1), dim3(256), 4096, 0, dsrc, ddst, n, doffs);
  std::vector<int> out(n / 4);
  hipMemcpy(out.data(), ddst, n, hipMemcpyDeviceToHost);
  int zeros = 0, stale = 0, good = 0, other = 0;
  for (int t = 0; t < 256; ++t) {
    for (int e = 0; e < 4; ++e) {
      const int v = out[t * 4 + e];
      if (t % 3 == 0) { if (v == 0x1000 + t * 4 + e) ++good; else ++other; }
      else { if (v == 0) ++zeros; else if (v == 0x7f7f7f7f) ++stale; else ++other; }
    }
  }
  printf("in-range dwords correct: %d / %d ; out-of-range lanes: %d dwords zero, %d dwords stale (sentinel kept), %d other\n", good, 4 * 86, zeros, stale, other);
  printf("RESULT: out-of-range LDS-DMA lanes %s\n", stale == 0 && other == 0 ? "WRITE ZEROS" : (zeros == 0 ? "ARE DROPPED (LDS keeps old data)" : "MIXED"));
  return 0;
}
