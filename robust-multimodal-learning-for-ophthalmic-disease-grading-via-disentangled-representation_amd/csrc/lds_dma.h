// Device helpers of the LDS-DMA kernels (conv_bf16_v3.hip, conv_wgrad_bf16_v3.hip): raw buffer descriptors as SGPR quads and the
// hand-issued buffer_load ... lds piece.
#pragma once
#include <hip/hip_runtime.h>

typedef int v3_i32x4 __attribute__((ext_vector_type(4)));
// Raw buffer descriptor (base, stride 0, num_records = bytes, gfx9 dword3 0x00020000) from wave-uniform values, as an SGPR quad
// for inline asm.
__device__ __forceinline__ v3_i32x4 v3_make_srd(const void* base, unsigned bytes) {
  const unsigned long a = (unsigned long)base;
  v3_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;
  return r;
}
// One LDS-DMA piece (64 lanes x 16 bytes -> 1 KiB of LDS at `lds`, wave-uniform) issued from inline asm: hipcc does not see it, so
// it neither counts it in vmcnt nor inserts its own conservative "LDS write pending" waits in front of ds_reads when the loop
// body has control flow (the builtin form made it drain vmcnt inside the staggered loop); every wait is placed by hand below.
// M0 carries the LDS destination and is written in the same statement (cdna_hip_programming.md 5.7); it is declared clobbered so
// that the compiler never keeps a value of its own in M0 across the statement.
__device__ __forceinline__ void v3_dma16(unsigned lds, unsigned voff, v3_i32x4 srd, int soff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds), "v"(voff), "s"(srd), "s"(soff) : "memory", "m0");
}

