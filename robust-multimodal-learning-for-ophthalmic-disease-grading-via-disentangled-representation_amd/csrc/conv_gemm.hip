// Implicit-GEMM convolution / linear kernels for fp32 tensors on the gfx950 matrix cores.
//
// Three contractions cover every conv and Linear on the EDRL hot path
// (reference call sites
// fusion_net.py:82-90,635-643,716-717,801-805 and the absent encoders :884-885):
//
//   gather  (fwd / dgrad):  D[m][n] = sum_k  Agather[m][k] * Wm[n][k]
//       m = destination pixel (NHWC row), k = (kh,kw,c_src), n = destination channel.
//       fwd  : src pixel = dst*stride - pad + tap
//       dgrad: src pixel = (dst + pad - tap)/stride when divisible
//       A Linear is the 1x1 case (OH=OW=1, rows = tokens).
//   wgrad  (TN):            dW[co][k] = sum_pix dY[pix][co] * Xcol[pix][k]   (split-K over pixels)
//
// Matrix instruction.  Shipped (EDRL_F32_SPLIT, below): v_mfma_f32_32x32x16_bf16 over an EXACT three-way bf16 split of both fp32
// operands, six products per fp32 product, fp32 accumulation -- the bf16 pipe is 16x the fp32 pipe on this chip.  With
// -DEDRL_F32_SPLIT=0 (libedrl_hip_f32mfma.so) and on the generic (non-FAST) paths: v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD).
// Block = 256 threads = 2x2 waves; wave tile (BM/2)x(BN/2) of 32x32 MFMA tiles.
// fp32-MFMA form: K-contiguous operands are staged to LDS as [row][16] (XOR-swizzled 16-byte chunks) and read with ds_read_b128:
// one read feeds 4 MFMA k-steps, lane half h taking k = 8*kc + 4*h + j; row-contiguous operands (wgrad) as [k][row], read with
// ds_read_b32.  Split form: three bf16 planes per operand, see SPL / SPLW in the kernels.
// Global->LDS is register staged and double buffered (one barrier per K tile): the next tiles' loads are issued before the
// running tile's MFMAs and written after them.
#include "edrl_common.h"
#include <atomic>
#include <mutex>
#include "edrl_config.h"
#include <type_traits>
#include <stdlib.h>
#include <string.h>

#define BK 32
#define LDK (BK + 4)

#include "conv_geom.h"
#ifndef EDRL_F32_SWZ
#define EDRL_F32_SWZ 1      // 0: the padded [row][BKT + 4] K-loop image of rounds 1-4 (A/B builds: make CXXFLAGS+=-DEDRL_F32_SWZ=0)
#endif
// EDRL_F32_SPLIT: fp32 contractions on the bf16 matrix pipe.  Each fp32 operand element is split (round to nearest) into three
// bf16 values a = a0 + a1 + a2 EXACTLY (8 + 8 + 8 mantissa bits with signed remainders), and a*b is formed as the six products
// a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0) on v_mfma_f32_32x32x16_bf16 with fp32 accumulation: every product is exact in
// fp32, the three dropped ones (a1b2, a2b1, a2b2) are below 2^-25 |ab| -- under the rounding of an fp32 multiply -- and the
// accumulator sees 6 roundings per 16 k where the fp32 MFMA path (v_mfma_f32_32x32x2_f32) sees 8.  Why: gfx950 runs the bf16
// MFMA at 16x the rate of the fp32 MFMA (2.5 PFLOP/s against 157 TFLOP/s), so six bf16 products cost 192 matrix-pipe cycles per
// 16 k of a 32 x 32 tile against 512 -- and the bf16 MFMA co-executes with the VALU, which the fp32 MFMA does not.
// Not reproduced: an operand element with |a| >= 2^128 (1 - 2^-9) (rounds to a bf16 infinity) or +-inf gives NaN where the fp32
// MFMA gives +-inf; below |a| ~ 2^-110 the middle / low planes are bf16 denormals (the element is then good to its high plane,
// 2^-9 relative, where the matrix pipe flushes them).  tests/test_gpu_kernels.py: exact reconstruction over 28 decades, exact
// integer results, error against fp64 next to the fp32-MFMA build's.
#ifndef EDRL_F32_SPLIT
#define EDRL_F32_SPLIT 1      // 0: fp32 MFMA (libedrl_hip_f32mfma.so is this file compiled with -DEDRL_F32_SPLIT=0)
#endif
#ifndef EDRL_F32_SPLIT_ATR2_LEAN
#define EDRL_F32_SPLIT_ATR2_LEAN 1  // 1: the ATR 2 variants without the accumulating epilogue run a register-lean K loop at 3 per CU
#endif
#ifndef EDRL_F32_SPLIT_OCC_EPI2
#define EDRL_F32_SPLIT_OCC_EPI2 3   // ... of the plain-operand data gradient with the sign-byte epilogue
#endif
#ifndef EDRL_F32_SPLIT_OCC_ATR2
#define EDRL_F32_SPLIT_OCC_ATR2 2   // ... of the data-gradient variants that form d_raw in the operand load (200-245 registers)
#endif
#ifndef EDRL_F32_SPLIT_OCC
#define EDRL_F32_SPLIT_OCC 3    // workgroups per CU of the split kernels (48 KiB of LDS each at 128 x 128)
#endif
// workgroups per CU of a gather-kernel variant: the split K loop keeps two register sets of operand loads in flight
constexpr int edrl_gather_occ(int bkt, bool fast, bool buf, int atr, int occ, int epi = 0) {
  if (!(EDRL_F32_SPLIT != 0 && bkt == 16 && fast && buf)) return occ;
  const int cap = atr == 2 ? ((EDRL_F32_SPLIT_ATR2_LEAN != 0 && epi != 2) ? 3 : EDRL_F32_SPLIT_OCC_ATR2)
                           : (epi == 2 ? EDRL_F32_SPLIT_OCC_EPI2 : EDRL_F32_SPLIT_OCC);
  return occ > cap ? cap : occ;
}
typedef __bf16 sp_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 sp_bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int sp_u32x2 __attribute__((ext_vector_type(2)));
typedef float sp_f32x4 __attribute__((ext_vector_type(4)));
// exact three-way split of four fp32 values into bf16 planes (round to nearest even at every level; remainders are exact).
// Per pair of elements and level: one v_cvt_pk_bf16_f32 and two v_dot2c_f32_bf16 -- the remainder a - hi comes straight from the
// PACKED pair, r = (hi0, hi1) . (-1, 0) + a0 (a dot product with a constant selects and widens the half in one instruction; the
// sum -hi + a is exact in fp32, so any rounding inside the dot product is a no-op): 14 vector instructions per 4 elements
// instead of 22 with an unpack + subtract per element.
typedef __bf16 sp_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void edrl_split3(sp_f32x4 v, sp_u32x2& p0, sp_u32x2& p1, sp_u32x2& p2) {
  // The selectors (-1, 0) / (0, -1) must reach the instruction as register operands: written as constants the compiler folds
  // (-1, 0) into the inline operand -1.0, which the hardware reads as the packed pair (0, -1) (scripts/microbench/
  // dot2_bf16_probe.hip: a0 - h1 comes back) -- so they are materialised through an opaque move.
  unsigned s0u = 0x0000BF80u, s1u = 0xBF800000u;
  asm volatile("" : "+s"(s0u), "+s"(s1u));
  const sp_bf16x2 sel0 = __builtin_bit_cast(sp_bf16x2, s0u), sel1 = __builtin_bit_cast(sp_bf16x2, s1u);
  sp_bf16x2 h[2], m[2], l[2];
  sp_f32x4 r, r2;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    h[q][0] = (__bf16)v[2 * q]; h[q][1] = (__bf16)v[2 * q + 1];
    r[2 * q] = __builtin_amdgcn_fdot2_f32_bf16(h[q], sel0, v[2 * q], false);
    r[2 * q + 1] = __builtin_amdgcn_fdot2_f32_bf16(h[q], sel1, v[2 * q + 1], false);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    m[q][0] = (__bf16)r[2 * q]; m[q][1] = (__bf16)r[2 * q + 1];
    r2[2 * q] = __builtin_amdgcn_fdot2_f32_bf16(m[q], sel0, r[2 * q], false);
    r2[2 * q + 1] = __builtin_amdgcn_fdot2_f32_bf16(m[q], sel1, r[2 * q + 1], false);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) { l[q][0] = (__bf16)r2[2 * q]; l[q][1] = (__bf16)r2[2 * q + 1]; }
  p0[0] = __builtin_bit_cast(unsigned, h[0]); p0[1] = __builtin_bit_cast(unsigned, h[1]);
  p1[0] = __builtin_bit_cast(unsigned, m[0]); p1[1] = __builtin_bit_cast(unsigned, m[1]);
  p2[0] = __builtin_bit_cast(unsigned, l[0]); p2[1] = __builtin_bit_cast(unsigned, l[1]);
}
// Weight-gradient image of the split path: per plane [16 pixels][W channels] bf16, unpadded; the 64-byte granule index of a row
// is XORed with a function of the pixel row so that the four rows of a transposing-read block (ds_read_b64_tr_b16: 4 pixels x 16
// channels per 16-lane group, two groups side by side) fall on four different 64-byte bank ranges, and so do the two pixel rows a
// 32-lane store pass touches.
template <int W>
__device__ __forceinline__ int edrl_wsplit_off(int row, int col) {          // byte offset of (pixel row, channel col) in a plane
  static_assert(W == 64 || W == 128, "tile widths of the weight-gradient kernel");
  const int gsw = W == 128 ? ((((row & 1) << 1) | ((row >> 1) & 1))) : ((row >> 1) & 1);
  return row * (W * 2) + (((col >> 5) ^ gsw) << 6) + ((col & 31) << 1);
}
typedef short sp_s16x4 __attribute__((ext_vector_type(4)));
// 8-deep fragment (pixels pix0 .. pix0 + 7 of channel col0 + (lane & 15)) of one plane: two transposing reads
template <int W>
__device__ __forceinline__ sp_bf16x8 edrl_wsplit_frag(const char* plane, int pix0, int col0, int lane) {
  const int g16 = lane & 15, q = g16 >> 2, p4 = g16 & 3;
  const char* a0 = plane + edrl_wsplit_off<W>(pix0 + q, col0 + 4 * p4);
  const sp_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sp_s16x4*)(a0));
  const sp_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sp_s16x4*)(a0 + 4 * (W * 2)));
  union { struct { sp_s16x4 l, h; } s; sp_bf16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
}
// (the variant with both operand transforms needs 190 registers: at 3 per CU it spills and runs 14 % slower than at 2,
// profiles/r05_f32_split_occ_ab.txt)
constexpr int edrl_wgrad_occ(int bkt, bool fastld, int occ, int xt) {
  if (!(EDRL_F32_SPLIT != 0 && bkt == 16 && fastld)) return occ;
  const int cap = xt != 0 ? 2 : EDRL_F32_SPLIT_OCC;
  return occ > cap ? cap : occ;
}

template <int BM, int BN, bool DGRAD, bool VEC>
__global__ __launch_bounds__(256, 2) void conv_gather_f32_kernel(
    const float* __restrict__ src, const float* __restrict__ wm, float* __restrict__ dst,
    const float* __restrict__ bias, const float* __restrict__ mul, GatherGeom g, int tiles_n) {
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_LD = (BM * BK / 4) / 256;  // float4 loads per thread
  constexpr int B_LD = (BN * BK / 4) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                   // [2][BM][LDK]
  float* Bs = smem + 2 * BM * LDK;    // [2][BN][LDK]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;

  const int nwg = gridDim.x;
  const int lid = edrl_xcd_remap(blockIdx.x, nwg);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  // ---- per-thread fixed staging coordinates
  const int k4 = (tid & 7) * 4;
  const int r0 = tid >> 3;  // rows r0 + 32*i
  int rbase[A_LD];          // image index * SH*SW  (pixel units), -1 if row invalid
  int rh[A_LD], rw[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const long m = m0 + r0 + 32 * i;
    if (m < g.M) {
      const int ohw = g.OHs * g.OWs;
      const int n = (int)(m / ohw);
      const int rem = (int)(m - (long)n * ohw);
      const int ii = rem / g.OWs, jj = rem - ii * g.OWs;
      const int oh = g.h0 + ii * g.step, ow = g.w0 + jj * g.step;
      rbase[i] = n;
      if (DGRAD) { rh[i] = oh + g.pad; rw[i] = ow + g.pad; }
      else       { rh[i] = oh * g.stride - g.pad; rw[i] = ow * g.stride - g.pad; }
    } else { rbase[i] = -1; rh[i] = 0; rw[i] = 0; }
  }

  f32x4 a_st[A_LD], b_st[B_LD];

  auto load_tile = [&](int kt) {
    const int k = kt * BK + k4;
    if (VEC) {
      const bool kvalid = k < g.Ktot;
      int tap = 0, c = 0, kh = 0, kw = 0;
      if (kvalid) {
        tap = k / g.SC; c = k - tap * g.SC;
        const int a = tap / g.KWs, b = tap - a * g.KWs;
        kh = g.kh0 + a * g.kstep; kw = g.kw0 + b * g.kstep;
      }
      const long woff = (long)(kh * g.KW + kw) * g.SC + c;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        bool ok = kvalid && rbase[i] >= 0;
        int sh, sw;
        if (DGRAD) {
          const int th = rh[i] - kh, tw = rw[i] - kw;   // divisible by the stride by construction of the class
          ok = ok && th >= 0 && tw >= 0;
          sh = th >> g.sshift; sw = tw >> g.sshift;
        } else { sh = rh[i] + kh; sw = rw[i] + kw; }
        ok = ok && sh >= 0 && sh < g.SH && sw >= 0 && sw < g.SW;
        if (ok) {
          const long pix = ((long)rbase[i] * g.SH + sh) * g.SW + sw;
          v = *reinterpret_cast<const f32x4*>(src + pix * g.ld_src + c);
        }
        a_st[i] = v;
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        const int n = n0 + r0 + 32 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (kvalid && n < g.NC) v = *reinterpret_cast<const f32x4*>(wm + (long)n * g.Kfull + woff);
        b_st[i] = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) a_st[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ke = k + e;
        const bool kvalid = ke < g.Ktot;
        int tap = 0, c = 0, kh = 0, kw = 0;
        if (kvalid) {
          tap = ke / g.SC; c = ke - tap * g.SC;
          const int a = tap / g.KWs, b = tap - a * g.KWs;
          kh = g.kh0 + a * g.kstep; kw = g.kw0 + b * g.kstep;
        }
        const long woff = (long)(kh * g.KW + kw) * g.SC + c;
#pragma unroll
        for (int i = 0; i < A_LD; ++i) {
          bool ok = kvalid && rbase[i] >= 0;
          int sh, sw;
          if (DGRAD) {
            const int th = rh[i] - kh, tw = rw[i] - kw;
            ok = ok && th >= 0 && tw >= 0;
            sh = th >> g.sshift; sw = tw >> g.sshift;
          } else { sh = rh[i] + kh; sw = rw[i] + kw; }
          ok = ok && sh >= 0 && sh < g.SH && sw >= 0 && sw < g.SW;
          float v = 0.f;
          if (ok) {
            const long pix = ((long)rbase[i] * g.SH + sh) * g.SW + sw;
            v = src[pix * g.ld_src + c];
          }
          a_st[i][e] = v;
        }
#pragma unroll
        for (int i = 0; i < B_LD; ++i) {
          const int n = n0 + r0 + 32 * i;
          float v = 0.f;
          if (kvalid && n < g.NC) v = wm[(long)n * g.Kfull + woff];
          b_st[i][e] = v;
        }
      }
    }
  };
  auto store_tile = [&](int buf) {
    float* a = As + buf * BM * LDK;
    float* b = Bs + buf * BN * LDK;
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      *reinterpret_cast<f32x4*>(a + (r0 + 32 * i) * LDK + k4) = a_st[i];
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      *reinterpret_cast<f32x4*>(b + (r0 + 32 * i) * LDK + k4) = b_st[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = (g.Ktot + BK - 1) / BK;
  load_tile(0);
  store_tile(0);
  __syncthreads();

  const int li = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < KT) load_tile(kt + 1);
    const float* a = As + buf * BM * LDK + (wm0 + li) * LDK + 4 * lh;
    const float* b = Bs + buf * BN * LDK + (wn0 + li) * LDK + 4 * lh;
#pragma unroll
    for (int kc = 0; kc < BK / 8; ++kc) {
      f32x4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDK + kc * 8);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDK + kc * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < KT) store_tile(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  const bool relu = g.flags & GF_RELU, accum = g.flags & GF_ACCUM;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + li;
    if (n >= g.NC) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < g.M) {
          long pix = m;
          if (DGRAD && g.step > 1) {   // sub-lattice rows -> destination pixel
            const int ohw = g.OHs * g.OWs;
            const int nn = (int)(m / ohw);
            const int rem = (int)(m - (long)nn * ohw);
            const int ii = rem / g.OWs, jj = rem - ii * g.OWs;
            pix = ((long)nn * g.OH + g.h0 + ii * g.step) * g.OW + g.w0 + jj * g.step;
          }
          float v = acc[i][j][r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          if (mul) v *= mul[pix * g.ld_aux + n];
          float* p = dst + pix * g.ld_dst + n;
          if (accum) v += *p;
          *p = v;
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Vectorised fast path (SC % 4 == 0): same tiling and LDS image as above, but the operand staging is
// branch-free (clamped address + select, so the 8 global loads of a K tile issue back to back) and is
// issued in four pieces, one per 8-deep MFMA chunk, so that its address arithmetic and the loads sit in
// the shadow of the 64-cycle fp32 MFMAs instead of in front of them.  The (tap, channel) decode of the
// K index advances incrementally (no integer division in the loop).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// 16-byte buffer store whose row advance rides in the scalar offset (no VALU per store).  gfx950 keeps reading the four data
// registers for a few cycles after issue; the compiler's hazard table covers that only for the immediate-offset form, and with
// an SGPR offset it schedules a VALU write to the first data register directly behind the store (observed: the written value
// lands in memory for part of the wave).  The store and the wait states are therefore one asm block.
__device__ __forceinline__ u32x4 edrl_rsrc_words(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  u32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32) & 0xffffu);
  r[2] = __builtin_amdgcn_readfirstlane(bytes);
  r[3] = 0x00020000u;
  return r;
}
__device__ __forceinline__ void edrl_buffer_store_b128_soff(f32x4 v, u32x4 rs, unsigned voff, int soff) {
  asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 7" : : "v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 g_bf16x4 __attribute__((ext_vector_type(4)));
// the same for 4 bf16 values (OUT16: the tile is stored rounded to bf16)
__device__ __forceinline__ void edrl_buffer_store_b64_soff(u32x2 v, u32x4 rs, unsigned voff, int soff) {
  asm volatile("buffer_store_dwordx2 %0, %1, %2, %3 offen\n\ts_nop 7" : : "v"(v), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
__device__ __forceinline__ u32x2 edrl_pack_bf16x4(f32x4 v) {
  g_bf16x4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
  return __builtin_bit_cast(u32x2, o);
}

// BUF (FAST only; host-checked footprints < 2 GiB): both operands come through buffer descriptors -- the gathered one
// through a per-workgroup descriptor based at the first image the tile's rows touch -- so masked rows / taps are an
// out-of-range 32-bit offset that the range check zero-fills: no 64-bit address arithmetic, no selects on the address
// or on the data, one add per staged piece and tile.
//
// Fused BatchNorm (BUF only; conv_geom.h GatherFuse):
//   ATR 1: the gathered operand is a RAW conv output and the activation relu((x-mean)*scale+shift) is formed while the
//          tile is staged to LDS (padding taps / rows >= M stay exactly 0) -- the producing layer's BatchNorm-apply pass
//          and its activated copy never exist.
//   ATR 2: the gathered operand is the BatchNorm-backward result d_raw = A*g - K1 - K2*(x - mean) formed from the masked
//          upstream gradient g (src) and the raw conv output x (src2) -- no bn_bwd_apply pass, d_raw never stored.
//   EPI 1: (data gradient) the tile just computed is the gradient of a BatchNorm+ReLU output: the epilogue masks it with
//          the ReLU decision (sign bytes, or recomputed from the raw tensor), stores the masked gradient and emits the
//          per-tile partial sums (sum g, sum g*xhat) of that BatchNorm's backward -- no separate reduction pass.
// MASK (ATR != 0): the geometry has padding taps / masked rows that must read as exactly 0 AFTER the transform (false for
// 1x1 / pad-0 layers, whose only invalid rows are rows >= M of the last tile: those are never stored).
// OUT16 (forward, EPI 0, vector epilogue, no accumulate / multiplier): `dst` is a bf16 tensor -- the fp32 result is rounded once
// on the way out (BatchNorm partials still come from the fp32 accumulators).  The bf16 trunk's stem: fp32 image in, fp32 MFMA,
// bf16 raw tensor out like every other layer of that trunk.
// VOL (BUF, ATR 0, EPI 0, MASK): 3-D convolution -- a third tap level (depth) in the row / tap decode, see GatherGeom.  The data
// gradient reaches it through edrl_conv3d_ndhwc_dgrad_f32, which hands every depth parity class to the kernel as a stride-1 depth
// geometry over a depth-reversed, class-compacted weight matrix (no kernel code of its own).
template <int BM, int BN, bool DGRAD, int BKT, int OCC, bool FAST, bool BUF = false, int ATR = 0, int EPI = 0, bool MASK = true,
          bool OUT16 = false, bool VOL = false>
__global__ __launch_bounds__(256, edrl_gather_occ(BKT, FAST, BUF, ATR, OCC, EPI))
void conv_gather_f32_v2_kernel(
    const float* __restrict__ src, const float* __restrict__ wm, float* __restrict__ dst,
    const float* __restrict__ bias, const float* __restrict__ mul, GatherGeom g, int tiles_n, GatherSplit S, GatherFuse F) {
  static_assert(ATR == 0 || BUF, "operand transforms ride on the buffer-descriptor path");
  static_assert(!OUT16 || (EPI == 0 && !DGRAD), "bf16 output: plain forward only");
  static_assert(!VOL || (BUF && FAST && ATR == 0 && EPI == 0 && MASK), "depth taps: plain forward / data gradient on the descriptor path");
  constexpr unsigned EB = OUT16 ? 2u : 4u;     // bytes per destination element
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  // K-loop LDS image (round 5, EDRL_F32_SWZ): rows of exactly BKT floats (64 bytes at BKT = 16), the 16-byte chunk index XORed
  // with (row >> 2) & 3.  Chosen against the hardware's lane groups (MI355X_MICROARCH.md, LDS): a ds_read_b128 group is 16 lanes =
  // rows {0-3, 12-15, 20-27} (or {4-11, 16-19, 28-31}) of a wave's 32 fragment rows, four of which share each 64-byte bank
  // quarter (row * 64 B mod 256 B) -- those four have four different (row >> 2) & 3, so the XOR spreads them over the quarter's
  // four slots: conflict-free; a ds_write_b128 group is 8 lanes = two whole rows = 128 contiguous bytes: conflict-free.  The
  // [row][BKT + 4] image of rounds 1-4 was conflict-free on the reads but 2-way on one slot of every store group
  // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.30 over the C1 step, profiles/r04_pmc_traffic_c1.json).
  constexpr bool SWZ = EDRL_F32_SWZ != 0 && BKT == 16 && FAST;
  // SPL (EDRL_F32_SPLIT, top of the file): the K-loop image holds three bf16 planes per operand, [plane][row][16 k] with 32-byte
  // rows; the 16-byte half of a row (k 0-7 / 8-15) sits at half ^ ((row >> 4) & 1): the ds_read_b128 lane groups (rows {0-3,
  // 12-15, 20-27} / {4-11, 16-19, 28-31} of one half) then cover the sixteen 16-byte slots of the 256-byte bank window once.
  constexpr bool SPL = EDRL_F32_SPLIT != 0 && BKT == 16 && FAST && BUF;
  constexpr int LDKT = SWZ ? BKT : BKT + 4;
  constexpr int KQ = BKT / 4;            // float4 columns per row of a K tile
  constexpr int RPP = 256 / KQ;          // rows covered by one staging piece
  constexpr int A_LD = BM / RPP;
  constexpr int B_LD = BN / RPP;
  static_assert(A_LD == BKT / 8, "one staging piece per 8-deep MFMA chunk");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * BM * LDKT;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  // K-split tail (GatherSplit, gather_ksplit_plan): workgroups past S.n_body each run 1/ksplit of the K loop of a tail tile and
  // leave their accumulators in S.slab (part >= 0); the fix-up launch (mode 2) sums the slabs in part order and runs the epilogue.
  int lid, part = -1;
  if (S.mode == 1 && (int)blockIdx.x >= S.n_body) {
    const int sub = (int)blockIdx.x - S.n_body;
    lid = S.n_body + sub / S.ksplit;
    part = sub - (sub / S.ksplit) * S.ksplit;
  } else if (S.mode == 2) {
    lid = S.n_body + (int)blockIdx.x;
  } else {
    lid = edrl_xcd_remap(blockIdx.x, S.mode == 1 ? (unsigned)S.n_body : gridDim.x);
  }
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  const int k4 = (tid % KQ) * 4;
  const int r0 = tid / KQ;
  int rn[A_LD], rh[A_LD], rw[A_LD];
  int rd[VOL ? A_LD : 1];                 // VOL: source depth of the row at depth tap 0 (do * dstride - dpad); rn = its source image
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const long m = m0 + r0 + RPP * i;
    if (m < g.M) {
      const int ohw = g.OHs * g.OWs;
      const int n = (int)(((unsigned long long)(unsigned)m * g.mg_ohw) >> g.sh_ohw);      // m / ohw (GatherGeom: magic division)
      const int rem = (int)m - n * ohw;
      const int ii = (int)(((unsigned long long)(unsigned)rem * g.mg_ow) >> g.sh_ow);     // rem / OWs
      const int jj = rem - ii * g.OWs;
      const int oh = g.h0 + ii * g.step, ow = g.w0 + jj * g.step;
      rn[i] = n;
      if constexpr (VOL) {                // n enumerates (sample, do): source image of depth tap kd = sample * SD + do * dstride - dpad + kd
        const int smp = (int)(((unsigned long long)(unsigned)n * g.mg_od) >> g.sh_od);
        rd[i] = (n - smp * g.OD) * g.dstride - g.dpad;
        rn[i] = smp * g.SD + rd[i];
      }
      if (DGRAD) { rh[i] = oh + g.pad; rw[i] = ow + g.pad; }
      else       { rh[i] = oh * g.stride - g.pad; rw[i] = ow * g.stride - g.pad; }
    } else { rn[i] = -1; rh[i] = 0; rw[i] = 0; if constexpr (VOL) rd[i] = -(1 << 29); }
  }
  long wrow[B_LD];
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int n = n0 + r0 + RPP * i;
    wrow[i] = n < g.NC ? (long)n * g.Kfull : -1;
  }

  // incremental decode of k = kt*BK + k4 -> (ta, tb, c): tap (kh0 + ta*kstep, kw0 + tb*kstep), channel c
  int c = k4, ta = 0, tb = 0, kk = k4;
  int td = 0;          // VOL: depth tap (innermost tap level)
  while (c >= g.SC) { c -= g.SC; if (++tb == g.KWs) { tb = 0; ++ta; } }

  // FAST (SC % BKT == 0): a K tile never straddles a tap, so every thread of the workgroup changes tap on the
  // same tile.  The per-row source offsets and the weight tap offset are then recomputed only on a tap change
  // (every SC/BKT tiles; never for 1x1 convs and Linear layers) and a staging piece costs ~one add + select.
  long rowoff[A_LD];   // element offset of the row's source pixel for the current tap, -1 = masked
  long tapoff = 0;     // (kh*KW + kw)*SC of the current tap
  // ---- BUF state: byte offsets into the two descriptors; >= 2 GiB = masked (stays masked under the per-tile adds)
  constexpr unsigned OOB = 0x80000000u;
  unsigned aoff[A_LD], boff[B_LD], wrow4[B_LD];
  __amdgpu_buffer_rsrc_t rs_a, rs_b, rs_a2, rs_p;
  int n_first = 0;
  int cb = 0;          // uniform part of c (c = cb + k4 while FAST)
  if constexpr (BUF) {
    const int ohw = g.OHs * g.OWs;
    long mlast = m0 + BM; if (mlast > g.M) mlast = g.M;
    n_first = (int)(((unsigned long long)(unsigned)m0 * g.mg_ohw) >> g.sh_ohw);
    int n_last = (int)(((unsigned long long)(unsigned)(mlast - 1) * g.mg_ohw) >> g.sh_ohw);
    if constexpr (VOL) {                  // descriptor over the whole volumes of the samples the tile's rows belong to
      n_first = (int)(((unsigned long long)(unsigned)n_first * g.mg_od) >> g.sh_od) * g.SD;
      n_last = (int)(((unsigned long long)(unsigned)n_last * g.mg_od) >> g.sh_od) * g.SD + g.SD - 1;
    }
    const unsigned a_bytes = (unsigned)(((long)(n_last - n_first + 1) * g.SH * g.SW - 1) * g.ld_src * 4 + (long)g.SC * 4);
    rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)n_first * g.SH * g.SW * g.ld_src), 0, (int)a_bytes, 0x00020000);
    rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wm, 0, (int)((long)g.NC * g.Kfull * 4), 0x00020000);
    if constexpr (ATR == 2)
      rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)F.src2 + (long)n_first * g.SH * g.SW * g.ld_src), 0, (int)a_bytes, 0x00020000);
    if constexpr (ATR != 0)
      rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)F.acoef, 0, 5 * g.SC * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int n = n0 + r0 + RPP * i;
      wrow4[i] = n < g.NC ? (unsigned)n * (unsigned)g.Kfull * 4u : OOB;
    }
  }
  auto retap = [&]() {
    const int kh = g.kh0 + ta * g.kstep, kw = g.kw0 + tb * g.kstep;
    tapoff = VOL ? (long)((kh * g.KW + kw) * g.KD + td) * g.SC : (long)(kh * g.KW + kw) * g.SC;
    if constexpr (BUF) {
      const bool kvalid = ta < g.KHs && g.KWs > 0;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        int sh, sw;
        bool ok = kvalid && (VOL ? (unsigned)(rd[VOL ? i : 0] + td) < (unsigned)g.SD : rn[i] >= 0);
        if (DGRAD) {
          const int th = rh[i] - kh, tw = rw[i] - kw;
          ok = ok && th >= 0 && tw >= 0;
          sh = th >> g.sshift; sw = tw >> g.sshift;
        } else { sh = rh[i] + kh; sw = rw[i] + kw; }
        ok = ok && (unsigned)sh < (unsigned)g.SH && (unsigned)sw < (unsigned)g.SW;
        const unsigned pix = (unsigned)(((rn[i] + (VOL ? td : 0) - n_first) * g.SH + sh) * g.SW + sw);
        aoff[i] = ok ? pix * (unsigned)(g.ld_src * 4) + (unsigned)(cb + k4) * 4u : OOB;
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i)
        boff[i] = kvalid ? wrow4[i] + (unsigned)((int)tapoff + cb + k4) * 4u : OOB;
      return;
    }
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      int sh, sw;
      bool ok = rn[i] >= 0;
      if (DGRAD) {
        const int th = rh[i] - kh, tw = rw[i] - kw;
        ok = ok && th >= 0 && tw >= 0;
        sh = th >> g.sshift; sw = tw >> g.sshift;
      } else { sh = rh[i] + kh; sw = rw[i] + kw; }
      ok = ok && sh >= 0 && sh < g.SH && sw >= 0 && sw < g.SW;
      rowoff[i] = ok ? (((long)rn[i] * g.SH + sh) * g.SW + sw) * g.ld_src : -1;
    }
  };
  const int KT = (g.Ktot + BKT - 1) / BKT;
  int kt0 = 0, kt1 = KT;
  if (S.mode == 2) kt1 = 0;                 // fix-up: no K loop, the accumulators come from the slabs
  if constexpr (BUF && FAST) {
    if (part >= 0) {                        // this workgroup's share of the K tiles; decode state of its first tile
      kt0 = part * S.kt_per;
      kt1 = kt0 + S.kt_per < KT ? kt0 + S.kt_per : KT;
      const int k0 = kt0 * BKT;
      int tq = k0 / g.SC;
      cb = k0 - tq * g.SC;
      if constexpr (VOL) { td = tq % g.KD; tq /= g.KD; }
      ta = tq / g.KWs; tb = tq - ta * g.KWs;
    }
  }
  if (FAST) retap();
  auto advance = [&]() {
    if constexpr (BUF) {   // the tap change is decided on the uniform part of the channel offset: a scalar branch
      cb += BKT;
      if (cb >= g.SC) {
        cb -= g.SC;
        if constexpr (VOL) { if (++td == g.KD) { td = 0; if (++tb == g.KWs) { tb = 0; ++ta; } } }
        else { if (++tb == g.KWs) { tb = 0; ++ta; } }
        retap();
      }
      else {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) aoff[i] += BKT * 4;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) boff[i] += BKT * 4;
      }
      return;
    }
    c += BKT; kk += BKT;
    if (FAST) {
      if (c >= g.SC) { c -= g.SC; if (++tb == g.KWs) { tb = 0; ++ta; } retap(); }
    } else {
      while (c >= g.SC) { c -= g.SC; if (++tb == g.KWs) { tb = 0; ++ta; } }
    }
  };

  f32x4 a_st[A_LD], b_st[B_LD];
  f32x4 a_st2[ATR == 2 ? A_LD : 1];      // ATR 2: the raw conv output paired with the gradient in a_st
  f32x4 tp0, tp1, tp2;                   // per-channel transform parameters of the staged K tile (channels cb+k4 .. +3)
  bool a_ok[A_LD], b_ok[B_LD];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load_piece = [&](int i) {
    if constexpr (BUF) {
      a_st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)aoff[i], 0, 0));
      if constexpr (ATR == 2) a_st2[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a2, (int)aoff[i], 0, 0));
      if constexpr (ATR != 0 && MASK) a_ok[i] = (int)aoff[i] >= 0;   // masked rows / padding taps / past-the-end tiles: offset >= 2 GiB
      if (i < B_LD) b_st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)boff[i], 0, 0));
      return;
    }
    const bool kvalid = kk < g.Ktot;   // (a class of the strided dgrad may have no taps at all: Ktot == 0)
    if (FAST) {   // raw load now, zero-select at store time (keeps the wait for the data off the MFMA chain's head)
      {
        const bool ok = kvalid && rowoff[i] >= 0;
        a_ok[i] = ok;
        a_st[i] = *reinterpret_cast<const f32x4*>(src + (ok ? rowoff[i] + c : 0));
      }
      if (i < B_LD) {
        const bool ok = kvalid && wrow[i] >= 0;
        b_ok[i] = ok;
        b_st[i] = *reinterpret_cast<const f32x4*>(wm + (ok ? wrow[i] + tapoff + c : 0));
      }
      return;
    }
    const int kh = g.kh0 + ta * g.kstep, kw = g.kw0 + tb * g.kstep;
    {
      int sh, sw;
      bool ok = kvalid && rn[i] >= 0;
      if (DGRAD) {
        const int th = rh[i] - kh, tw = rw[i] - kw;
        ok = ok && th >= 0 && tw >= 0;
        sh = th >> g.sshift; sw = tw >> g.sshift;
      } else { sh = rh[i] + kh; sw = rw[i] + kw; }
      ok = ok && sh >= 0 && sh < g.SH && sw >= 0 && sw < g.SW;
      const long pix = ((long)rn[i] * g.SH + sh) * g.SW + sw;
      const long off = ok ? pix * g.ld_src + c : 0;
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + off);
      a_st[i] = ok ? v : zero4;
    }
    if (i < B_LD) {
      const bool ok = kvalid && wrow[i] >= 0;
      const long off = ok ? wrow[i] + (long)(kh * g.KW + kw) * g.SC + c : 0;
      const f32x4 v = *reinterpret_cast<const f32x4*>(wm + off);
      b_st[i] = ok ? v : zero4;
    }
  };
  // The per-channel parameters of the staged tile are fetched late (before the LAST MFMA chunk of the running tile, ~1000
  // cycles ahead of their use in store_tile) so that they are not live across the whole chain (VGPR budget at 3 per CU).
  auto load_params = [&]() {
    if constexpr (ATR != 0) {
      const int po = (cb + k4) * 4;      // channels cb+k4 .. +3 (< SC always: cb < SC, SC % BKT == 0); rows via the scalar offset
      if constexpr (ATR == 1) {          // tp0 = scale (row 2), tp1 = shift2 (row 4)
        tp0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po, 2 * g.SC * 4, 0));
        tp1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po, 4 * g.SC * 4, 0));
      } else {                           // tp0 = A (row 0), tp1 = nK2 (row 1), tp2 = C2 (row 2)
        tp0 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po, 0, 0));
        tp1 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po, g.SC * 4, 0));
        tp2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po, 2 * g.SC * 4, 0));
      }
    }
  };
  // The operand transform runs on the staged registers, one 4-channel piece at a time (two packed FMAs [+ max]); in the K loop
  // the pieces are issued behind the last MFMAs of the running tile.
  auto transform_piece = [&](int i) {     // i compile-time after unrolling
    f32x4 v;
    if constexpr (ATR == 1) v = edrl_bn_relu2(a_st[i], tp0, tp1);
    else if constexpr (ATR == 2) v = edrl_bn_bwd_dx2(a_st[i], a_st2[i], tp0, tp1, tp2);
    else v = a_st[i];
    if constexpr (ATR != 0 && MASK) v = a_ok[i] ? v : zero4;
    a_st[i] = v;
  };
  auto transform_tile = [&]() {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) transform_piece(i);
  };
  auto store_tile = [&](int buf) {
    float* a = As + buf * BM * LDKT;
    float* b = Bs + buf * BN * LDKT;
    // (SWZ: RPP = 64 rows per piece, so (row >> 2) & 3 of a thread's rows r0 + 64 i is (r0 >> 2) & 3 for every piece)
    const int k4s = SWZ ? (k4 ^ (((r0 >> 2) & 3) << 2)) : k4;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const f32x4 v = (!FAST || BUF || a_ok[i]) ? a_st[i] : zero4;
      *reinterpret_cast<f32x4*>(a + (r0 + RPP * i) * LDKT + k4s) = v;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      *reinterpret_cast<f32x4*>(b + (r0 + RPP * i) * LDKT + k4s) = (!FAST || BUF || b_ok[i]) ? b_st[i] : zero4;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  if constexpr (SPL) {
    // Split K loop.  The matrix-pipe chain of a K tile is 24 MFMAs x 32 cycles -- shorter than a memory latency -- so the operand
    // loads run TWO tiles ahead in two register sets: iteration t issues the loads of tile t + 2 into set t & 1, multiplies tile t
    // out of LDS buffer t & 1, and (transforms,) splits and stores tile t + 1 from set (t + 1) & 1 into the other buffer.
    f32x4 qa[2][A_LD], qb[2][B_LD], qa2[ATR == 2 ? 2 : 1][A_LD];
    bool qok[2][A_LD];
    auto sp_load = [&](auto PC) {
      constexpr int P = decltype(PC)::value;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        qa[P][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)aoff[i], 0, 0));
        if constexpr (ATR == 2) qa2[P][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a2, (int)aoff[i], 0, 0));
        if constexpr (ATR != 0 && MASK) qok[P][i] = (int)aoff[i] >= 0;
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) qb[P][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)boff[i], 0, 0));
    };
    const int sq = tid % KQ;
    const int sso = r0 * 32 + ((((sq >> 1) ^ (r0 >> 4)) & 1) << 4) + ((sq & 1) << 3);   // (rows r0 + 64 i: bit 4 of the row = bit 4 of r0)
    auto sp_finish = [&](auto PC, int buf) {
      constexpr int P = decltype(PC)::value;
      char* a = (char*)smem + buf * (BM * 96) + sso;
      char* b = (char*)smem + 2 * BM * 96 + buf * (BN * 96) + sso;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        f32x4 v = qa[P][i];
        if constexpr (ATR == 1) v = edrl_bn_relu2(v, tp0, tp1);
        if constexpr (ATR == 2) v = edrl_bn_bwd_dx2(v, qa2[ATR == 2 ? P : 0][i], tp0, tp1, tp2);
        if constexpr (ATR != 0 && MASK) v = qok[P][i] ? v : zero4;
        sp_u32x2 p0, p1, p2;
        edrl_split3(v, p0, p1, p2);
        char* d = a + RPP * i * 32;
        *reinterpret_cast<sp_u32x2*>(d) = p0;
        *reinterpret_cast<sp_u32x2*>(d + BM * 32) = p1;
        *reinterpret_cast<sp_u32x2*>(d + 2 * BM * 32) = p2;
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        sp_u32x2 p0, p1, p2;
        edrl_split3(qb[P][i], p0, p1, p2);
        char* d = b + RPP * i * 32;
        *reinterpret_cast<sp_u32x2*>(d) = p0;
        *reinterpret_cast<sp_u32x2*>(d + BN * 32) = p1;
        *reinterpret_cast<sp_u32x2*>(d + 2 * BN * 32) = p2;
      }
    };
    const int hs = ((lh ^ (li >> 4)) & 1) << 4;
    auto sp_iter = [&](auto PC) {
      constexpr int P = decltype(PC)::value;
      load_params();                    // of tile t + 1 (the decode state is one tile ahead of the LDS image)
      advance();                        // -> tile t + 2
      sp_load(PC);
      __builtin_amdgcn_sched_barrier(0);
      const char* pa = (const char*)smem + P * (BM * 96) + (wm0 + li) * 32 + hs;
      const char* pb = (const char*)smem + 2 * BM * 96 + P * (BN * 96) + (wn0 + li) * 32 + hs;
      sp_bf16x8 fa[3][TM], fb[2][TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const sp_bf16x8*>(pb + j * 1024);
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const sp_bf16x8*>(pa + p * (BM * 32) + i * 1024);
      // weight plane q meets pixel planes 0 .. 2 - q: 3 + 2 + 1 = 6 products per element pair
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const int cur = q & 1, nxt = cur ^ 1;
        if (q + 1 < 3) {
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[nxt][j] = *reinterpret_cast<const sp_bf16x8*>(pb + (q + 1) * (BN * 32) + j * 1024);
        }
#pragma unroll
        for (int p = 2 - q; p >= 0; --p)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[p][i], fb[cur][j], acc[i][j], 0, 0, 0);
      }
      sp_finish(std::integral_constant<int, P ^ 1>{}, P ^ 1);
      __syncthreads();
    };
    if constexpr (EDRL_F32_SPLIT_ATR2_LEAN != 0 && ATR == 2 && EPI != 2) {
      // Register-lean form for the variants that form d_raw in the operand load: ONE register set (loads one tile ahead), weight
      // fragments single-buffered -- 150 registers instead of 201-207, so 3 workgroups per CU instead of 2: -2 % over the
      // ResNet-50 layers, -5 % on the 1x1 layers of stages 3-4 (profiles/r05_f32_split_occ_ab.txt).  The accumulating epilogue
      // (EPI 2) spills at 3 per CU and loses 12 %: it keeps the two-set loop at 2.
      const std::integral_constant<int, 0> S0{};
      sp_load(S0);
      load_params();
      sp_finish(S0, 0);
      __syncthreads();
      for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        advance();
        load_params();
        sp_load(S0);
        __builtin_amdgcn_sched_barrier(0);
        const char* pa = (const char*)smem + buf * (BM * 96) + (wm0 + li) * 32 + hs;
        const char* pb = (const char*)smem + 2 * BM * 96 + buf * (BN * 96) + (wn0 + li) * 32 + hs;
        sp_bf16x8 fa[3][TM], fb[TN];
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[p][i] = *reinterpret_cast<const sp_bf16x8*>(pa + p * (BM * 32) + i * 1024);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const sp_bf16x8*>(pb + q * (BN * 32) + j * 1024);
#pragma unroll
          for (int p = 2 - q; p >= 0; --p)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[p][i], fb[j], acc[i][j], 0, 0, 0);
        }
        sp_finish(S0, buf ^ 1);
        __syncthreads();
      }
    } else {
    sp_load(std::integral_constant<int, 0>{});
    load_params();
    sp_finish(std::integral_constant<int, 0>{}, 0);
    __syncthreads();
    advance();
    sp_load(std::integral_constant<int, 1>{});
    for (int kt = kt0; kt < kt1; kt += 2) {
      sp_iter(std::integral_constant<int, 0>{});
      if (kt + 1 < kt1) sp_iter(std::integral_constant<int, 1>{});
    }
    }
  } else {
#pragma unroll
  for (int i = 0; i < A_LD; ++i) load_piece(i);
  load_params();
  transform_tile();
  store_tile(0);
  __syncthreads();
  }

  for (int kt = kt0; kt < (SPL ? kt0 : kt1); ++kt) {
    const int buf = (kt - kt0) & 1;
    advance();   // decode state of tile kt+1 (past the end: kvalid is false and the pieces load zeros)
    // (SWZ: fragment rows wm0 + li + 32 i have (row >> 2) & 3 = (li >> 2) & 3; chunk lh + 2 kc of the row sits at chunk ^ that)
    const int ch0 = SWZ ? (lh ^ ((li >> 2) & 3)) : lh;
    const float* a = As + buf * BM * LDKT + (wm0 + li) * LDKT + 4 * ch0;
    const float* b = Bs + buf * BN * LDKT + (wn0 + li) * LDKT + 4 * ch0;
    const int kc1 = SWZ ? ((ch0 ^ 2) - ch0) * 4 : 8;     // float offset of chunk lh + 2 (the second 8-deep half) from chunk lh
    // The next tile's global loads are issued FIRST (pinned with a scheduling barrier): they then have the whole
    // tile's MFMA chain (>= 2048 cycles) to land before the ds_write at the bottom.  Left to itself the compiler
    // sinks them to the end of the chain (to shorten live ranges) and the wave stalls on vmcnt every tile.
    if (FAST) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) load_piece(i);
      if constexpr (ATR == 1) load_params();       // (ATR 2: fetched before the last chunk, VGPR budget)
      __builtin_amdgcn_sched_barrier(0);
    }
    // fragment registers are double buffered: chunk kc+1's LDS reads are issued ahead of chunk kc's 16 MFMAs
    f32x4 af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDKT);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDKT);
#pragma unroll
    for (int kc = 0; kc < BKT / 8; ++kc) {
      const int cur = kc & 1, nxt = cur ^ 1;
      if (kc + 1 < BKT / 8) {
#pragma unroll
        for (int i = 0; i < TM; ++i) af[nxt][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * LDKT + (SWZ ? kc1 : (kc + 1) * 8));
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[nxt][j] = *reinterpret_cast<const f32x4*>(b + j * 32 * LDKT + (SWZ ? kc1 : (kc + 1) * 8));
      }
      if (!FAST) load_piece(kc);
      if constexpr (ATR == 2) {
        if (kc == BKT / 8 - 1) { __builtin_amdgcn_sched_barrier(0); load_params(); __builtin_amdgcn_sched_barrier(0); }
      }
      if constexpr (ATR != 0) {
        if (kc == BKT / 8 - 1) {
          // last chunk: the operand transform of the NEXT tile is issued element by element behind the MFMAs of the
          // chunk's second half (each slice pinned behind its MFMA), so its VALU instructions execute in the matrix
          // pipe's shadow instead of between the end of the chain and the workgroup barrier.
          constexpr int NM = 4 * TM * TN;
#pragma unroll
          for (int q = 0; q < NM; ++q) {
            const int s = q / (TM * TN), i = (q / TN) % TM, j = q % TN;
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][s], bf[cur][j][s], acc[i][j], 0, 0, 0);
            if (q >= NM - A_LD) {          // one piece behind each of the last A_LD MFMAs
              __builtin_amdgcn_sched_barrier(0);
              transform_piece(q - (NM - A_LD));
              __builtin_amdgcn_sched_barrier(0);
            }
          }
          continue;
        }
      }
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i][s], bf[cur][j][s], acc[i][j], 0, 0, 0);
    }
    if (FAST) __builtin_amdgcn_sched_barrier(0);   // the zero-selects + ds_writes (and their vmcnt wait) stay below the chain
    store_tile(buf ^ 1);
    __syncthreads();
  }

  if (S.mode != 0) {
    constexpr int NE = TM * TN * 16;         // accumulator floats per lane
    if (part >= 0) {                         // partial K range: accumulators -> slab [(tail tile, part)][element][thread]
      float* sl = S.slab + ((long)((int)blockIdx.x - S.n_body) * NE) * 256 + tid;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) sl[((i * TN + j) * 16 + r) * 256] = acc[i][j][r];
      return;
    }
    if (S.mode == 2) {                       // fix-up: parts summed in part order (fixed order: deterministic)
      for (int p = 0; p < S.ksplit; ++p) {
        const float* sl = S.slab + ((long)((int)blockIdx.x * S.ksplit + p) * NE) * 256 + tid;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += sl[((i * TN + j) * 16 + r) * 256];
      }
    }
  }
  const bool relu = g.flags & GF_RELU, accum = g.flags & GF_ACCUM;
  if (g.flags & GF_VEC_EPI) {
    // Vector epilogue: each wave transposes its accumulators through LDS (the K-loop buffers are free now) so that a
    // lane owns 4 consecutive channels of one pixel: 16-B stores, 16 lanes per 256-B row segment, instead of 64
    // scalar stores per lane.  Bias / ReLU / mask / accumulate are applied on the float4.
    // Staging row stride = WN floats, NO padding: the hardware's ds_read_b128 lane groups ({0-3,12-15,20-27}, ...:
    // MI355X_MICROARCH.md, LDS) are built so that a LINEAR image with 256-byte (64-float) or 128-byte rows is conflict-free, and
    // the ds_write_b32 transposing stores bank per 32-lane half (consecutive floats: conflict-free at any stride).  The WN + 4
    // padding of rounds 1-3 (a CUDA habit) put row r + 1's float4 slots onto row r's banks for two lanes of every group:
    // 32 % of this kernel's LDS-active cycles were bank conflicts (profiles/r03_pmc_traffic_c1.json).
    constexpr int SLD = WN;
    constexpr int C4 = WN / 4;             // float4 per staged row
    constexpr int RPP2 = 64 / C4;          // rows per pass of the wave
    float* stage = smem + wave * 32 * SLD;
    const int srow = lane / C4, sc4 = lane % C4;
    const int n = n0 + wn0 + sc4 * 4;
    f32x4 bv4 = {0.f, 0.f, 0.f, 0.f};
    if (bias && n < g.NC) bv4 = *reinterpret_cast<const f32x4*>(bias + n);
    const bool stats = EPI == 0 && (g.flags & GF_STATS) != 0;
    f32x4 kshift = {0.f, 0.f, 0.f, 0.f}, st0 = {0.f, 0.f, 0.f, 0.f}, st1 = {0.f, 0.f, 0.f, 0.f};
    // EPI 1: BatchNorm(+ReLU) backward of the tensor this tile is the gradient of (channels n .. n+3 of this lane)
    // e_mean: the partial sum of plane 1 is taken SHIFTED, sum g*(x - mean) (the batch mean is known here, unlike in the forward
    // statistics), so that the fp64 finalize has no mean*sum(g) to cancel: |mean|/sigma no longer amplifies the fp32 rounding
    f32x4 e_scale = zero4, e_shift2 = zero4, e_mean = zero4;
    if constexpr (EPI != 0) {
      if (n < g.NC) e_mean = *reinterpret_cast<const f32x4*>(F.ep_fcoef + n);
      if (n < g.NC && !F.ep_mask) {
        e_scale = *reinterpret_cast<const f32x4*>(F.ep_fcoef + 2 * (long)g.NC + n);
        e_shift2 = *reinterpret_cast<const f32x4*>(F.ep_fcoef + 4 * (long)g.NC + n);
      }
    }
    // Lean path (every row of the tile valid, no bias / ReLU / multiplier, destination pixel = row): next to the fp32 MFMA
    // every VALU instruction is matrix-pipe time taken from the other workgroups of the CU, so the stores (and the EPI operand
    // loads) go through per-wave buffer descriptors -- lane offset fixed, the row advance in the scalar offset, the column
    // guard an out-of-range lane offset -- the loads of a 32-row pass are all issued before its LDS transpose, and the only
    // vector work per row group is the arithmetic itself.  Specialised (no per-row-group branches):
    //   EPI 0: plain store (+ BatchNorm statistics), no accumulate
    //   EPI 1: ReLU decision recomputed from ep_x (GF_EPI_RELU), no accumulate      (data gradients inside a block)
    //   EPI 2: sign bytes, accumulate into dst                                     (block-input gradient)
    // every other combination takes the general loop below.
    const bool lean_flags = EPI == 0 ? !accum
                          : EPI == 1 ? (!accum && !F.ep_mask && (g.flags & GF_EPI_RELU))
                                     : (F.ep_mask != nullptr);       // (EPI 2: sign bytes; accumulate or not -- a uniform branch)
    const bool lean = m0 + BM <= g.M && !bias && !mul && !relu && lean_flags &&
                      (!(DGRAD && g.step > 1) || (g.flags & GF_LEAN_STRIDED));
    if (lean) {
      constexpr unsigned OOBE = 0x80000000u;
      constexpr int NT = 32 / RPP2;                         // row groups per 32-row pass
      const bool colok = n < g.NC;
      const int nq = g.NC >> 2;
      // STRIDED (parity class of a strided data gradient): the rows of the tile are scattered destination pixels, so the lane
      // offset of each row group is decoded (magic division) relative to a per-workgroup descriptor based at the first image
      // the tile touches, and the scalar offset is 0.  Otherwise: per-wave descriptor at the wave's first row, fixed lane
      // offset, the row advance in the scalar offset.
      auto run = [&](auto STR) {
        constexpr bool S = decltype(STR)::value;
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int wrow0 = (wv >> 1) * WM;
        long pix0; int colb; unsigned span;                  // descriptor base (destination pixel, column) and its rows
        if constexpr (S) {
          const int nf = (int)(((unsigned long long)(unsigned)m0 * g.mg_ohw) >> g.sh_ohw);
          const int nl = (int)(((unsigned long long)(unsigned)(m0 + BM - 1) * g.mg_ohw) >> g.sh_ohw);
          pix0 = (long)nf * g.OH * g.OW; colb = 0; span = (unsigned)(nl - nf + 1) * (unsigned)(g.OH * g.OW);
        } else { pix0 = m0 + wrow0; colb = n0 + (wv & 1) * WN; span = WM; }
        void* dbase = OUT16 ? (void*)((__bf16*)dst + pix0 * g.ld_dst + colb) : (void*)(dst + pix0 * g.ld_dst + colb);
        const u32x4 ws_d = edrl_rsrc_words(dbase, span * (unsigned)g.ld_dst * EB);
        const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(dbase, 0, (int)(span * (unsigned)g.ld_dst * EB), 0x00020000);
        __amdgpu_buffer_rsrc_t rs_x, rs_k;
        if constexpr (EPI != 0)
          rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)((const float*)F.ep_x + pix0 * F.ld_ep + colb), 0,
                                                   (int)(span * (unsigned)F.ld_ep * 4u), 0x00020000);
        if constexpr (EPI == 2)
          rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)(F.ep_mask + pix0 * nq + (colb >> 2)), 0, (int)(span * (unsigned)nq), 0x00020000);
        const int ldd4 = (int)g.ld_dst * (int)EB, ldx4 = (int)F.ld_ep * 4;
        // lane offsets (non-strided: fixed; strided: column part here, pixel part per row group)
        const unsigned cd = colok ? (unsigned)((S ? n : sc4 * 4) * 4) : OOBE;
        const unsigned cdd = colok ? (unsigned)(S ? n : sc4 * 4) * EB : OOBE;     // (destination: EB bytes per element)
        const unsigned ck = colok ? (unsigned)(S ? (n >> 2) : sc4) : OOBE;
        const unsigned vd = cdd + (S ? 0u : (unsigned)(srow * ldd4));
        const unsigned vx = cd + (S ? 0u : (unsigned)(srow * ldx4));
        const unsigned vk = ck + (S ? 0u : (unsigned)(srow * nq));
        f32x4 xr[EPI != 0 ? NT : 1], old[EPI == 2 ? NT : 1];
        int kb[EPI == 2 ? NT : 1];
        int prel[S ? NT : 1];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          if constexpr (S) {      // destination pixel of each row group of the pass, relative to pix0
            const int ohw = g.OHs * g.OWs;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const int m = (int)m0 + wm0 + i * 32 + t * RPP2 + srow;
              const int nn = (int)(((unsigned long long)(unsigned)m * g.mg_ohw) >> g.sh_ohw);
              const int rem = m - nn * ohw;
              const int ii = (int)(((unsigned long long)(unsigned)rem * g.mg_ow) >> g.sh_ow);
              const int jj = rem - ii * g.OWs;
              prel[t] = (int)(((long)nn * g.OH + g.h0 + ii * g.step) * g.OW + g.w0 + jj * g.step - pix0);
            }
          }
          auto od = [&](int t) { return S ? vd + (unsigned)(prel[S ? t : 0] * ldd4) : vd; };
          auto ox = [&](int t) { return S ? vx + (unsigned)(prel[S ? t : 0] * ldx4) : vx; };
          auto ok = [&](int t) { return S ? vk + (unsigned)(prel[S ? t : 0] * nq) : vk; };
          auto so = [&](int t, int step) { return S ? 0 : (i * NT + t) * step; };
          if constexpr (EPI != 0) {       // all operand loads of the pass in flight together: one memory latency per pass
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              xr[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)ox(t), so(t, RPP2 * ldx4), 0));
              if constexpr (EPI == 2) {
                if (accum) old[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_d, (int)od(t), so(t, RPP2 * ldd4), 0));
                kb[t] = __builtin_amdgcn_raw_buffer_load_b8(rs_k, (int)ok(t), so(t, RPP2 * nq), 0);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + j * 32 + li] = acc[i][j][r];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          if (stats && i == 0) kshift = *reinterpret_cast<const f32x4*>(stage + sc4 * 4);
          if (EPI == 0 && stats) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(stage + (t * RPP2 + srow) * SLD + sc4 * 4);
              const f32x4 d = v - kshift;
              st0 += d;
              st1 = __builtin_elementwise_fma(d, d, st1);
              if constexpr (OUT16) edrl_buffer_store_b64_soff(edrl_pack_bf16x4(v), ws_d, od(t), so(t, RPP2 * ldd4));
              else edrl_buffer_store_b128_soff(v, ws_d, od(t), so(t, RPP2 * ldd4));
            }
          } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
              f32x4 v = *reinterpret_cast<const f32x4*>(stage + (t * RPP2 + srow) * SLD + sc4 * 4);
              if constexpr (EPI == 1) {
                const f32x4 pre = edrl_bn_pre2(xr[t], e_scale, e_shift2);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pre[e] > 0.f ? v[e] : 0.f;
              }
              if constexpr (EPI == 2) {
                if (accum) v += old[t];
#pragma unroll
                for (int e = 0; e < 4; ++e) {    // bit e of the sign byte -> all-ones / zero (1-bit signed field extract), AND
                  const float ve = v[e];         // (a copy: bit_cast applied to the vector element itself reads element 0)
                  v[e] = __int_as_float(__float_as_int(ve) & ((kb[t] << (31 - e)) >> 31));
                }
              }
              if constexpr (EPI != 0) {
                st0 += v;
                st1 = __builtin_elementwise_fma(v, xr[t] - e_mean, st1);
              }
              if constexpr (OUT16) edrl_buffer_store_b64_soff(edrl_pack_bf16x4(v), ws_d, od(t), so(t, RPP2 * ldd4));
              else edrl_buffer_store_b128_soff(v, ws_d, od(t), so(t, RPP2 * ldd4));
            }
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
      };
      if (DGRAD && g.step > 1) run(std::true_type{}); else run(std::false_type{});
    } else
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + j * 32 + li] = acc[i][j][r];
      // each wave transposes through its OWN staging region: a wave-local fence suffices (no workgroup barrier)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      // shift K of the BatchNorm partials = the WAVE's first output row (row 0 of its own staging region): a sample of
      // the data, so var << mean^2 costs no precision.  The two row halves of the tile are re-based onto the upper
      // half's K when they are combined below.
      if (stats && i == 0) kshift = *reinterpret_cast<const f32x4*>(stage + sc4 * 4);
#pragma unroll
      for (int t = 0; t < 32 / RPP2; ++t) {
        const int row = t * RPP2 + srow;
        const long m = m0 + wm0 + i * 32 + row;
        if (m < g.M && n < g.NC) {
          long pix = m;
          if (DGRAD && g.step > 1) {
            const int ohw = g.OHs * g.OWs;
            const int nn = (int)(m / ohw);
            const int rem = (int)(m - (long)nn * ohw);
            const int ii = rem / g.OWs, jj = rem - ii * g.OWs;
            pix = ((long)nn * g.OH + g.h0 + ii * g.step) * g.OW + g.w0 + jj * g.step;
          }
          f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * SLD + sc4 * 4) + bv4;
          if (stats) { const f32x4 d = v - kshift; st0 += d; st1 += d * d; }
          if (relu) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
          }
          if (mul) v *= *reinterpret_cast<const f32x4*>(mul + pix * g.ld_aux + n);
          if constexpr (OUT16) {      // (host-checked: no accumulate, no EPI)
            *reinterpret_cast<u32x2*>((__bf16*)dst + pix * g.ld_dst + n) = edrl_pack_bf16x4(v);
            continue;
          }
          float* p = dst + pix * g.ld_dst + n;
          if (accum) v += *reinterpret_cast<const f32x4*>(p);
          if constexpr (EPI != 0) {
            const f32x4 xr = *reinterpret_cast<const f32x4*>((const float*)F.ep_x + pix * F.ld_ep + n);
            if (F.ep_mask) {
              const int mb = F.ep_mask[pix * (long)(g.NC >> 2) + (n >> 2)];
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (mb >> e) & 1 ? v[e] : 0.f;
            } else if (g.flags & GF_EPI_RELU) {
              const f32x4 pre = edrl_bn_pre2(xr, e_scale, e_shift2);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = pre[e] > 0.f ? v[e] : 0.f;
            }
            st0 += v;                                    // sum g
            st1 = __builtin_elementwise_fma(v, xr - e_mean, st1); // sum g*(x - mean)  (x rstd = sum g*xhat in the fp64 finalize)
          }
          *reinterpret_cast<f32x4*>(p) = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // staging reads done before the next pass overwrites them
    }
    if constexpr (EPI != 0) {
      // (sum g, sum g*(x - mean)) of the tile's valid rows -> F.ep_part[chunk0 + tile_m][2][NC]
#pragma unroll
      for (int o = 32; o >= C4; o >>= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { st0[e] += __shfl_xor(st0[e], o, 64); st1[e] += __shfl_xor(st1[e], o, 64); }
      }
      float* red = smem + 4 * 32 * SLD;          // [wave][2][WN]
      if (srow == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[(wave * 2 + 0) * WN + sc4 * 4 + e] = st0[e];
          red[(wave * 2 + 1) * WN + sc4 * 4 + e] = st1[e];
        }
      }
      __syncthreads();
      if ((wave >> 1) == 0 && srow == 0 && n < g.NC) {
        float* pp = F.ep_part + ((long)F.ep_chunk0 + tile_m) * 2 * g.NC;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cc = sc4 * 4 + e;
          pp[n + e] = st0[e] + red[((wave + 2) * 2 + 0) * WN + cc];
          pp[g.NC + n + e] = st1[e] + red[((wave + 2) * 2 + 1) * WN + cc];
        }
      }
      return;
    }
    if (stats) {
      // lanes that share the channel group (same sc4) differ by multiples of C4: butterfly over those lane bits
#pragma unroll
      for (int o = 32; o >= C4; o >>= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { st0[e] += __shfl_xor(st0[e], o, 64); st1[e] += __shfl_xor(st1[e], o, 64); }
      }
      float* red = smem + 4 * 32 * SLD;          // beyond the staging regions: [wave][3][WN] = (S1, S2, K)
      if (srow == 0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[(wave * 3 + 0) * WN + sc4 * 4 + e] = st0[e];
          red[(wave * 3 + 1) * WN + sc4 * 4 + e] = st1[e];
          red[(wave * 3 + 2) * WN + sc4 * 4 + e] = kshift[e];
        }
      }
      __syncthreads();
      if ((wave >> 1) == 0 && srow == 0 && n < g.NC) {   // waves 0/1 own the column halves; add the lower row half (waves 2/3)
        float* pp = g.stat_part + (long)tile_m * 3 * g.NC;
        long nl = g.M - (m0 + WM);                       // valid rows of the lower half
        const float nb = nl <= 0 ? 0.f : (nl > WM ? (float)WM : (float)nl);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cc = sc4 * 4 + e;
          const float s1b = red[((wave + 2) * 3 + 0) * WN + cc], s2b = red[((wave + 2) * 3 + 1) * WN + cc];
          const float d = red[((wave + 2) * 3 + 2) * WN + cc] - kshift[e];      // K_lower - K_upper
          // sum (y-Ku) = sum (y-Kl) + n d ;  sum (y-Ku)^2 = sum (y-Kl)^2 + 2 d sum (y-Kl) + n d^2
          pp[n + e] = st0[e] + (s1b + nb * d);
          pp[g.NC + n + e] = st1[e] + (s2b + 2.f * d * s1b + nb * d * d);
          pp[2 * g.NC + n + e] = kshift[e];
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + li;
    if (n >= g.NC) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m < g.M) {
          long pix = m;
          if (DGRAD && g.step > 1) {
            const int ohw = g.OHs * g.OWs;
            const int nn = (int)(m / ohw);
            const int rem = (int)(m - (long)nn * ohw);
            const int ii = rem / g.OWs, jj = rem - ii * g.OWs;
            pix = ((long)nn * g.OH + g.h0 + ii * g.step) * g.OW + g.w0 + jj * g.step;
          }
          float v = acc[i][j][r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          if (mul) v *= mul[pix * g.ld_aux + n];
          float* p = dst + pix * g.ld_dst + n;
          if (accum) v += *p;
          *p = v;
        }
      }
    }
  }
}

// kernels of the gather family issued by this process (edrl_gather_launch_count): a call may issue several (parity classes of a
// strided data gradient, body + tail of a split call); the bench's timer reads the difference around a call so that its launch
// count is the one rocprofv3 sees
static std::atomic<long> g_gather_launches{0};      // (launchers run on autograd worker threads as well as the main thread)

template <int BM, int BN, bool DGRAD, int BKT, int OCC, bool FAST, bool BUF = false, int ATR = 0, int EPI = 0, bool MASK = true,
          bool OUT16 = false, bool VOL = false>
static int launch_gather_v2(const float* src, const float* wm, float* dst, const float* bias,
                            const float* mul, const GatherGeom& g, hipStream_t st, const GatherFuse* fuse = nullptr,
                            const GatherSplit* split = nullptr) {
  const int tiles_m = edrl_cdiv(g.M, BM), tiles_n = edrl_cdiv(g.NC, BN);
  GatherSplit S;
  memset(&S, 0, sizeof(S));
  long nblk = (long)tiles_m * tiles_n;
  if (split) {
    S = *split;
    const long tail = nblk - S.n_body;      // tail tiles: K-split in the main launch (mode 1), finished by the fix-up launch (mode 2)
    nblk = S.mode == 1 ? S.n_body + tail * S.ksplit : tail;
  }
  if (nblk <= 0) return 0;
  if (nblk > 0x7fffffffL) return EDRL_EINVAL;
  constexpr bool SPL = EDRL_F32_SPLIT != 0 && BKT == 16 && FAST && BUF;
  const size_t lds = SPL ? (size_t)2 * (BM + BN) * 96 : (size_t)2 * (BM + BN) * (BKT + 4) * sizeof(float);
  auto kern = conv_gather_f32_v2_kernel<BM, BN, DGRAD, BKT, OCC, FAST, BUF, ATR, EPI, MASK, OUT16, VOL>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  GatherFuse F;
  if (fuse) F = *fuse; else memset(&F, 0, sizeof(F));
  GatherGeom gm = g;
  gather_geom_magic(&gm);
  if constexpr (VOL) gather_magic(g.OD > 0 ? (unsigned)g.OD : 1u, &gm.mg_od, &gm.sh_od);
  {   // strided parity class: the lean epilogue addresses the destination (and the EPI operands) through a descriptor based
      // at the first image a tile touches
    const long ohw = (long)g.OHs * g.OWs;
    const long ldmax = (fuse && fuse->ld_ep > g.ld_dst) ? fuse->ld_ep : g.ld_dst;
    if (DGRAD && g.step > 1 && ohw > 0 && (BM / ohw + 2) * g.OH * g.OW * ldmax * 4 < (1L << 31)) gm.flags |= GF_LEAN_STRIDED;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, src, wm, dst, bias, mul, gm, tiles_n, S, F);
  ++g_gather_launches;
  EDRL_LAUNCH_CHECK();
  return 0;
}

// Partial last quantum.  With 4 (3) workgroups resident per CU a launch takes ceil(workgroups / 256) workgroup times on the busiest
// CU: 1568 tiles of 128 x 128 (l4 at 1024 images) cost as much as 1792 (scripts/tail_sweep.py: the time steps every 256
// workgroups, 137 TFLOP/s at exact multiples, 121 at 6.12 x 256).  When the tail beyond the last multiple of 256 is at most 128
// tiles, each tail tile's K loop is split over floor(256 / tail) workgroups appended to the SAME launch (they fill the last
// quantum with 1/ksplit of a tile's work each and leave their accumulators in a slab), and a second, short launch of the same
// kernel sums the slabs in a fixed order and runs the tile's ordinary epilogue (statistics, masks, accumulate: every variant).
// Deterministic; not bit-identical to the unsplit kernel on the tail tiles (the K sum is associated differently).
// Measured and dropped on the way: the tail as 128 x 64 tiles in a second launch (+1..2 % only: the body's ragged end and the
// launch boundary eat the gain; the narrow tile itself is 8-10 % slower).
// The slab the split workgroups leave their accumulators in is CALLER-OWNED (kernels never allocate): the
// host wrapper registers one per (device, stream) it launches the family on -- edrl_gather_ksplit_set_workspace, memory from the
// torch caching allocator (<package>/_lib.py) -- and the library only keeps the table.  No slab registered for a launch's (device,
// stream): no split (the unsplit kernel, bit-identical on the body tiles).  Launches of one stream are ordered, so one slab per
// stream suffices; g_slab_mutex additionally spans the main + fix-up launch pair, so two host threads that issue split calls on
// one stream cannot interleave their halves.
static const size_t GATHER_SLAB_BYTES = (size_t)256 * 128 * 128 * sizeof(float);     // 256 split workgroups x one 128 x 128 tile
static const int GATHER_SLAB_MAX = 64;
static struct { int dev; hipStream_t stream; float* ptr; } g_slab[GATHER_SLAB_MAX];
static int g_slab_n = 0;
static std::mutex g_slab_mutex;
static float* gather_slab_locked(hipStream_t st) {      // g_slab_mutex held
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  for (int i = 0; i < g_slab_n; ++i)
    if (g_slab[i].dev == dev && g_slab[i].stream == st) return g_slab[i].ptr;
  return nullptr;
}
static bool gather_ksplit_plan(const GatherGeom& g, GatherSplit* S, int bn) {
  if (!edrl_cfg().gather_tail_split) return false;
  const int tiles_m = edrl_cdiv(g.M, 128), tiles_n = edrl_cdiv(g.NC, bn);
  if (tiles_n <= 0 || tiles_n > 256 || (256 % tiles_n)) return false;
  const int per = 256 / tiles_n;                         // row tiles per 256 workgroups
  const int body = tiles_m / per * per;
  const long tail = (long)(tiles_m - body) * tiles_n;    // tail workgroups (body == 0: a grid that leaves half of the CUs idle)
  if (tail == 0 || tail > 128 || (body == 0 && edrl_cfg().gather_tail_split < 2)) return false;
  const int KT = edrl_cdiv(g.Ktot, 16);
  int ks = (int)(256 / tail);
  if (ks > 8) ks = 8;
  while (ks > 1 && edrl_cdiv(KT, ks) < 8) --ks;          // >= 8 K tiles per part
  if (ks < 2) return false;
  // worth it only when the saved part of a tile time ((1 - 1/ks) x ~61 ns per K element, measured) exceeds the fix-up launch (~30 us)
  if ((long)g.Ktot * (ks - 1) / ks < 640) return false;
  S->mode = 1; S->n_body = body * tiles_n; S->ksplit = ks; S->kt_per = edrl_cdiv(KT, ks); S->slab = nullptr;
  return true;
}

// One call = one launch, or (K-split tail) the main launch + its fix-up launch
template <int BN, bool DGRAD, int OCC, int ATR = 0, int EPI = 0, bool MASK = true>
static int launch_gather_split(const float* src, const float* wm, float* dst, const float* bias, const float* mul,
                               const GatherGeom& g, hipStream_t st, const GatherFuse* fuse = nullptr) {
  GatherSplit S;
  if (!gather_ksplit_plan(g, &S, BN))
    return launch_gather_v2<128, BN, DGRAD, 16, OCC, true, true, ATR, EPI, MASK>(src, wm, dst, bias, mul, g, st, fuse);
  std::lock_guard<std::mutex> lock(g_slab_mutex);       // held across the launch pair
  S.slab = gather_slab_locked(st);
  if (!S.slab)
    return launch_gather_v2<128, BN, DGRAD, 16, OCC, true, true, ATR, EPI, MASK>(src, wm, dst, bias, mul, g, st, fuse);
  for (int mode = 1; mode <= 2; ++mode) {
    S.mode = mode;
    const int rc = launch_gather_v2<128, BN, DGRAD, 16, OCC, true, true, ATR, EPI, MASK>(src, wm, dst, bias, mul, g, st, fuse, &S);
    if (rc) return rc;
  }
  return 0;
}

// The fused-BatchNorm variants exist on the buffer-descriptor fast path only: report whether a geometry qualifies.
static bool gather_fused_ok(const float* src, const float* wm, const float* dst, const GatherGeom& g) {
  const long ohw = (long)g.OHs * g.OWs;
  return (g.SC % 16 == 0) && (g.ld_src % 4 == 0) && (g.Kfull % 4 == 0) && (g.NC % 4 == 0) && (g.ld_dst % 4 == 0) &&
         ((((uintptr_t)src | (uintptr_t)wm | (uintptr_t)dst) & 15) == 0) && ohw > 0 &&
         (128 / ohw + 2) * g.SH * g.SW * g.ld_src * 4 < (1L << 31) && (long)g.NC * g.Kfull * 4 < (1L << 31);   // (rows < 2^31 is enforced by every extern "C" launcher before GatherGeom.M is formed)
}
// workgroups per CU: the forward operand transform fits the 128-VGPR budget of 4 per CU, the data-gradient variants (second
// operand tensor + epilogue reduction) need the 168 of 3 per CU
#define FUSED_OCC (ATR == 1 ? 4 : 3)
template <bool DGRAD, int ATR, int EPI>
static int dispatch_gather_fused(const float* src, const float* wm, float* dst, const GatherGeom& g0, const GatherFuse& F,
                                 hipStream_t st) {
  if (!gather_fused_ok(src, wm, dst, g0)) return EDRL_EINVAL;
  GatherGeom g = g0;
  g.flags |= GF_VEC_EPI;
  const int small_grid = edrl_cfg().narrow_below;
  const bool narrow = g.NC <= 64 || ((long)edrl_cdiv(g.M, 128) * edrl_cdiv(g.NC, 128) < small_grid);
  const bool mask = !(g.KH == 1 && g.KW == 1 && g.pad == 0);   // 1x1 / pad 0: no padding taps, no masked rows below M
  if (narrow) {
    if (mask) return launch_gather_split<64, DGRAD, FUSED_OCC, ATR, EPI, true>(src, wm, dst, nullptr, nullptr, g, st, &F);
    return launch_gather_split<64, DGRAD, FUSED_OCC, ATR, EPI, false>(src, wm, dst, nullptr, nullptr, g, st, &F);
  }
  if (mask) return launch_gather_split<128, DGRAD, FUSED_OCC, ATR, EPI, true>(src, wm, dst, nullptr, nullptr, g, st, &F);
  return launch_gather_split<128, DGRAD, FUSED_OCC, ATR, EPI, false>(src, wm, dst, nullptr, nullptr, g, st, &F);
}

template <int BM, int BN, bool DGRAD, bool VEC>
static int launch_gather(const float* src, const float* wm, float* dst, const float* bias,
                         const float* mul, const GatherGeom& g, hipStream_t st) {
  const int tiles_m = edrl_cdiv(g.M, BM), tiles_n = edrl_cdiv(g.NC, BN);
  const long nblk = (long)tiles_m * tiles_n;
  if (nblk <= 0) return 0;
  if (nblk > 0x7fffffffL) return EDRL_EINVAL;
  const size_t lds = (size_t)2 * (BM + BN) * LDK * sizeof(float);
  auto kern = conv_gather_f32_kernel<BM, BN, DGRAD, VEC>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, src, wm, dst, bias, mul, g, tiles_n);
  ++g_gather_launches;
  EDRL_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------ small-M linear (the head's batch-level projections)
// dst[m][n] = act(sum_k src[m][k] * wm[n][k] + bias[n]) * mul[m][n] (+ dst) for M <= 64 rows (B = 32 / 64 samples: the q-len-1/2
// attention blocks, PoE / guided projections of fusion_net.py:635-643, 555-566, 929-939): a weight-streaming GEMV-like shape that the
// 128-row implicit-GEMM tiles run at 1-3 TFLOP/s (16-48 workgroups, each walking all of K serially).  Here one workgroup owns 16
// output columns; its 4 waves split K four ways, every lane streams 32 contiguous bytes of a weight row per step straight into the
// fragment of v_mfma_f32_16x16x4_f32 (A = activations, B = weights: element e of the lane's two float4 is the k of MFMA e, the
// same on both operands, so any fixed assignment of k to lanes is a valid contraction order), and the four partial tiles are
// summed in wave order through LDS (deterministic).  N/16 workgroups x 4 waves stream the N x K weights exactly once.
template <int MT>
__global__ __launch_bounds__(256) void linear_smallm_f32_kernel(const float* __restrict__ src, const float* __restrict__ wm,
                                                                float* __restrict__ dst, const float* __restrict__ bias,
                                                                const float* __restrict__ mul, int M, int N, int K, long ld_src,
                                                                long ld_dst, long ld_aux, int flags) {
  __shared__ float red[4][MT][256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int n0 = blockIdx.x * 16;
  const int kper = K >> 2, kbeg = wave * kper;
  f32x4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float* wrow = wm + (long)(n0 + r) * K + 8 * q;
  const float* arow[MT];
  bool aok[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) { aok[t] = t * 16 + r < M; arow[t] = src + (long)(aok[t] ? t * 16 + r : 0) * ld_src + 8 * q; }
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int k0 = kbeg; k0 < kbeg + kper; k0 += 32) {
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(wrow + k0), b1 = *reinterpret_cast<const f32x4*>(wrow + k0 + 4);
    f32x4 a0[MT], a1[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      a0[t] = aok[t] ? *reinterpret_cast<const f32x4*>(arow[t] + k0) : z4;
      a1[t] = aok[t] ? *reinterpret_cast<const f32x4*>(arow[t] + k0 + 4) : z4;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[t][e], b0[e], acc[t], 0, 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[t][e], b1[e], acc[t], 0, 0, 0);
  }
  // C/D map of the 16x16 tile: col = lane&15 (n), row = 4*(lane>>4) + reg (m)
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[wave][t][e * 64 + lane] = acc[t][e];
  __syncthreads();
  const bool relu = flags & GF_RELU, accum = flags & GF_ACCUM;
  const int e = tid >> 6;                       // thread -> (reg e, lane)
  const int n = n0 + (lane & 15);
  const float bv = bias ? bias[n] : 0.f;
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int m = t * 16 + 4 * (lane >> 4) + e;
    if (m < M) {
      float v = ((red[0][t][tid] + red[1][t][tid]) + red[2][t][tid]) + red[3][t][tid] + bv;
      if (relu) v = fmaxf(v, 0.f);
      if (mul) v *= mul[(long)m * ld_aux + n];
      float* p = dst + (long)m * ld_dst + n;
      if (accum) v += *p;
      *p = v;
    }
  }
}

// Geometry test + launch; returns -1 when the small-M kernel does not apply
static int try_linear_smallm(const float* src, const float* wm, float* dst, const float* bias, const float* mul,
                             const GatherGeom& g, hipStream_t st) {
  const bool on = edrl_cfg().linear_smallm != 0;
  if (!on || g.KH != 1 || g.KW != 1 || g.pad != 0 || g.step != 1 || g.stride != 1 || g.OHs * g.OWs != 1 || g.SH * g.SW != 1) return -1;
  if (g.M > 64 || g.M <= 0 || (g.NC % 16) || (g.Ktot % 128) || g.Ktot != g.Kfull || (g.ld_src % 4) || (g.flags & ~(GF_RELU | GF_ACCUM)))
    return -1;
  if ((((uintptr_t)src | (uintptr_t)wm) & 15) || g.stat_part) return -1;
  const int mt = (g.M + 15) / 16;
  const dim3 grid(g.NC / 16), blk(256);
  if (mt <= 2)
    hipLaunchKernelGGL(linear_smallm_f32_kernel<2>, grid, blk, 0, st, src, wm, dst, bias, mul, g.M, g.NC, g.Ktot, g.ld_src, g.ld_dst, g.ld_aux, g.flags);
  else
    hipLaunchKernelGGL(linear_smallm_f32_kernel<4>, grid, blk, 0, st, src, wm, dst, bias, mul, g.M, g.NC, g.Ktot, g.ld_src, g.ld_dst, g.ld_aux, g.flags);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}

template <bool DGRAD>
static int dispatch_gather(const float* src, const float* wm, float* dst, const float* bias,
                           const float* mul, const GatherGeom& g, hipStream_t st) {
  {
    const int rc = try_linear_smallm(src, wm, dst, bias, mul, g, st);
    if (rc >= 0) return rc;
  }
  const bool vec = (g.SC % 4 == 0) && (g.ld_src % 4 == 0) && (g.Kfull % 4 == 0) &&
                   (((uintptr_t)src & 15) == 0) && (((uintptr_t)wm & 15) == 0);
  // 64-wide N tiles for Cout <= 64 and for small grids (the head's Linear layers at 32..1568 rows leave most of the
  // 256 CUs idle with 128-wide tiles: twice the workgroups, same work each K step)
  const int small_grid = edrl_cfg().narrow_below;
  const bool narrow = g.NC <= 64 || ((long)edrl_cdiv(g.M, 128) * edrl_cdiv(g.NC, 128) < small_grid);
  GatherGeom gv = g;
  if (vec && (g.NC % 4 == 0) && (g.ld_dst % 4 == 0) && (((uintptr_t)dst & 15) == 0) &&
      (!bias || ((uintptr_t)bias & 15) == 0) && (!mul || ((g.ld_aux % 4 == 0) && ((uintptr_t)mul & 15) == 0)))
    gv.flags |= GF_VEC_EPI;
  if (vec) {
    const GatherGeom& g = gv;
    // K tile 16 -> 40 KiB of LDS and 128 VGPRs per workgroup: 3 workgroups (12 waves) per CU.  Measured on the
    // ResNet-50 layer shapes (1024 images): +13 % over K tile 32 / 2 workgroups per CU (profiles/).
    const int variant = edrl_cfg().gather_variant;
    if (variant == 0) {   // K tile 32, 2 workgroups per CU (kept for A/B runs)
      if (narrow) return launch_gather_v2<128, 64, DGRAD, 32, 2, false>(src, wm, dst, bias, mul, g, st);
      return launch_gather_v2<128, 128, DGRAD, 32, 2, false>(src, wm, dst, bias, mul, g, st);
    }
    const bool fast = (g.SC % 16 == 0) && variant != 3;
    if (fast) {
      // buffer-descriptor path: 128 rows touch at most 128/(OHs*OWs) + 2 images; both footprints must fit 31 bits
      const bool buf_env = edrl_cfg().gather_buf != 0;
      const long ohw = (long)g.OHs * g.OWs;
      const bool buf = buf_env && ohw > 0 && (128 / ohw + 2) * g.SH * g.SW * g.ld_src * 4 < (1L << 31) &&
                       (long)g.NC * g.Kfull * 4 < (1L << 31);   // (rows < 2^31: checked by the extern "C" launchers)
      if (buf && variant != 5) {   // 4 workgroups per CU (123 VGPRs, 4 x 40 KiB = all of the LDS): +2-3 % over 3 per CU (variant 5)
        if (narrow) return launch_gather_split<64, DGRAD, 4>(src, wm, dst, bias, mul, g, st);
        return launch_gather_split<128, DGRAD, 4>(src, wm, dst, bias, mul, g, st);
      }
      if (buf) {
        if (narrow) return launch_gather_v2<128, 64, DGRAD, 16, 3, true, true>(src, wm, dst, bias, mul, g, st);
        return launch_gather_v2<128, 128, DGRAD, 16, 3, true, true>(src, wm, dst, bias, mul, g, st);
      }
      if (narrow) return launch_gather_v2<128, 64, DGRAD, 16, 3, true>(src, wm, dst, bias, mul, g, st);
      return launch_gather_v2<128, 128, DGRAD, 16, 3, true>(src, wm, dst, bias, mul, g, st);
    }
    if (narrow) return launch_gather_v2<128, 64, DGRAD, 16, 3, false>(src, wm, dst, bias, mul, g, st);
    return launch_gather_v2<128, 128, DGRAD, 16, 3, false>(src, wm, dst, bias, mul, g, st);
  }
  if (narrow) return launch_gather<128, 64, DGRAD, false>(src, wm, dst, bias, mul, g, st);
  return launch_gather<128, 128, DGRAD, false>(src, wm, dst, bias, mul, g, st);
}

// ------------------------------------------------------------------ wgrad (TN, split-K)
struct WgradGeom {
  long P;           // pixels = N*OH*OW (reduction length)
  int OH, OW;
  int Co;           // rows of dW
  int SH, SW, SC;
  int KH, KW, stride, pad;
  int Ktot;         // KH*KW*SC columns of dW
  long ld_dy, ld_x;
  int tiles_per_split;
  int tiles_x, tiles_y;   // dW tiles along K and Co: the grid is 1-D (tiles_x * tiles_y * splits), XCD-remapped
  // Depth taps (VOL instantiations only: 3-D convolution over NDHWC volumes, edrl_conv3d_ndhwc_wgrad_f32): the pixel index
  // enumerates (sample, do, oh, ow), the columns of dW run over (kh, kw, kd, ci) -- the weight layout of the depth-unfolded form --
  // and the im2col operand is read from the volume itself, source image sample*SD + do*dstride - dpad + kd.
  int KD, SD, OD, dstride, dpad;
};

// FASTLD (host-checked: VEC, 16/OW + 1 <= OH, per-block operand footprints < 2 GiB): both operands are fetched with
// buffer loads through per-block descriptors whose range check zero-fills every out-of-range row, so the loop carries
// no 64-bit address arithmetic, no zero-selects and no loops in the pixel decode: per staged row a 32-bit running
// offset advanced by constants (one tile = BKT output pixels) with two branch-free wrap corrections.
// Fused BatchNorm operands (FASTLD only; WgradFuse):
//   DYT 2: dY = A*g - K1 - K2*(yraw - mean) is formed from the masked gradient g (`dy`) and the raw conv output (F.dy2)
//          while the tile is staged (the BatchNorm-backward apply pass and the d_raw tensor never exist);
//   XT 1:  the im2col operand is relu((xraw - mean)*scale + shift) of the RAW previous conv output (`x`).
// Rows outside the split's pixel range / padding taps are forced to exactly 0 after the transform.
struct WgradFuse {
  const float* dy2;                 // DYT 2: raw conv output of this layer (same geometry as dy)
  const float* bcoef;               // DYT 2: [4][Co] = A, nK2, C2, mean
  const float* xcoef;               // XT 1:  [5][SC] = mean, rstd, scale, shift, shift2 of the producing BatchNorm
};
// MASKX (XT 1): the conv has padding taps, which must read as exactly 0 after the transform (3x3); a 1x1 / pad-0 layer's only
// invalid X rows are those past the end of the tensor, and they meet dY rows that DYT has already zeroed.
// DY16 (FASTLD, no transforms): `dy` is a bf16 tensor, widened exactly on load (the stem of the bf16 trunk: its d_raw is stored as bf16).
// VOL (FASTLD, no transforms): depth taps, see WgradGeom -- the thread's pixel-row state gains a third level (the source depth of
// tap 0 and its wrap at a sample boundary), the depth tap of a column is a per-thread constant like its (kh, kw).
template <int BM, int BN, bool VEC, int BKT, int OCC, bool FASTLD = false, int DYT = 0, int XT = 0, bool MASKX = true,
          bool DY16 = false, bool VOL = false>
__global__ __launch_bounds__(256, edrl_wgrad_occ(BKT, FASTLD, OCC, XT)) void conv_wgrad_f32_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part, WgradGeom g, WgradFuse F) {
  static_assert((DYT == 0 && XT == 0) || FASTLD, "operand transforms ride on the buffer-descriptor path");
  static_assert(!VOL || (FASTLD && DYT == 0 && XT == 0 && !DY16), "depth taps: plain buffer-load path only");
  static_assert(!DY16 || (FASTLD && DYT == 0), "bf16 dY: plain buffer-load path only");
  constexpr unsigned EBY = DY16 ? 2u : 4u;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_LD = (BM * BKT / 4) / 256, B_LD = (BN * BKT / 4) / 256;
  constexpr int AC4 = BM / 4, BC4 = BN / 4;
  constexpr int LDA = BM + 4, LDB = BN + 4;
  // SPLW (EDRL_F32_SPLIT, top of the file): three bf16 planes per operand, [buffer][plane][16 pixels][channels]
  constexpr bool SPLW = EDRL_F32_SPLIT != 0 && BKT == 16 && FASTLD;
  constexpr int PLA = 16 * BM * 2, PLB = 16 * BN * 2;      // bytes per plane
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                    // [2][BK][LDA]
  float* Bs = smem + 2 * BKT * LDA;     // [2][BK][LDB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  // XCD-aware order: the tiles_x*tiles_y workgroups of one split (they all stream the same dY / X pixel range) get
  // consecutive logical ids, and consecutive logical ids share an XCD, i.e. one L2 fetches that range once.
  const int lid = edrl_xcd_remap(blockIdx.x, gridDim.x);
  const int per_split = g.tiles_x * g.tiles_y;
  const int split = lid / per_split;
  const int trem = lid - split * per_split;
  const int tyi = trem / g.tiles_x;
  const int co0 = tyi * BM, n0 = (trem - tyi * g.tiles_x) * BN;

  const int ptiles = (int)((g.P + BKT - 1) / BKT);             // (P < 2^31: host-checked)
  const int t_begin = split * g.tiles_per_split;
  int t_end = t_begin + g.tiles_per_split;
  if (t_end > ptiles) t_end = ptiles;

  // fixed column decode for the im2col operand
  const int bc4 = tid % BC4;
  const int kcol = n0 + bc4 * 4;
  int ktap = 0, kc = 0, kkh = 0, kkw = 0;
  const bool kvalid = kcol < g.Ktot;
  if (VEC && kvalid) { ktap = kcol / g.SC; kc = kcol - ktap * g.SC; kkh = ktap / g.KW; kkw = ktap - kkh * g.KW; }
  const int ac4 = tid % AC4;
  const int ohw = g.OH * g.OW;

  f32x4 a_st[A_LD], b_st[B_LD];
  // Vector path: the (n, oh, ow) decode of each staged pixel row advances incrementally from tile to tile (BKT pixels
  // per tile; no division in the loop) and the loads are branch-free (clamped address + select) so that they issue
  // back to back ahead of the MFMA chain.
  int bn_[B_LD], boh[B_LD], bow[B_LD];
  long bp[B_LD], ap[A_LD];
  if (VEC) {
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const long p = (long)t_begin * BKT + (tid + 256 * i) / BC4;
      bp[i] = p;
      const long n = p / ohw;
      const int rem = (int)(p - n * ohw);
      bn_[i] = (int)n; boh[i] = rem / g.OW; bow[i] = rem - boh[i] * g.OW;
    }
#pragma unroll
    for (int i = 0; i < A_LD; ++i) ap[i] = (long)t_begin * BKT + (tid + 256 * i) / AC4;
  }
  // ---- FASTLD state
  constexpr unsigned OOB = 0x80000000u;         // offsets at/above 2 GiB stay outside every descriptor (ranges < 2 GiB)
  // FASTLD im2col mapping: a thread's B_LD float4 sit in ONE pixel row (row = tid / 16, columns (tid % 16 + 16 i) * 4), so the
  // (n, oh, ow) state of the row and its two wrap corrections are kept once per thread and only the tap part is per load.
  unsigned a_off[A_LD], b_roff = 0;
  int b_ih = 0, b_iw = 0;                       // oh*stride - pad, ow*stride - pad of the thread's pixel row
  int f_kh[B_LD], f_kw[B_LD], f_tapc[B_LD];     // tap of load i and its byte offset (incl. channel) relative to (b_ih, b_iw)
  int f_kd[VOL ? B_LD : 1];                     // VOL: depth tap of load i
  int b_id = 0, id_lim = 0, v_ods = 0, v_dstr = 0;   // VOL: do*dstride - dpad of the thread's pixel row, its wrap limit / amounts
  unsigned v_wrapd = 0;
  bool f_kval[B_LD];
  int ih_lim = 0, iw_lim = 0;
  unsigned a_step = 0, c_step = 0, c_wrapw = 0, c_wraph = 0;
  int dw_step = 0, dh_step = 0;
  __amdgpu_buffer_rsrc_t rs_dy, rs_x, rs_dy2;
  unsigned dy_last = 0, x_last = 0;            // last in-range 16-byte offset of each descriptor
  f32x4 q0 = {0.f, 0.f, 0.f, 0.f}, q1 = q0, q2 = q0;            // DYT 2 parameters (A, nK2, C2) of this thread's 4 output channels
  f32x4 xs[B_LD], xb[B_LD];                                     // XT 1 parameters (scale, shift2) of this thread's input channels
  bool a_ok[A_LD], b_ok[B_LD];
  f32x4 a2_st[DYT == 2 ? A_LD : 1];
  if constexpr (FASTLD) {
    const long p_lo = (long)t_begin * BKT;
    long p_hi = (long)t_end * BKT; if (p_hi > g.P) p_hi = g.P;
    long rows = p_hi - p_lo; if (rows < 1) rows = 1;
    const unsigned ld4y = (unsigned)g.ld_dy * EBY, ld4x = (unsigned)(g.ld_x * 4);
    const unsigned dy_bytes = (unsigned)((rows - 1) * ld4y + (unsigned)g.Co * EBY);
    rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)dy + p_lo * g.ld_dy * (long)EBY), 0, (int)dy_bytes, 0x00020000);
    dy_last = dy_bytes - 4u * EBY;
    if constexpr (DYT == 2) {
      rs_dy2 = __builtin_amdgcn_make_buffer_rsrc((void*)(F.dy2 + p_lo * g.ld_dy), 0, (int)dy_bytes, 0x00020000);
      const int co = co0 + ac4 * 4;
      if (co < g.Co) {
        q0 = *reinterpret_cast<const f32x4*>(F.bcoef + co);
        q1 = *reinterpret_cast<const f32x4*>(F.bcoef + g.Co + co);
        q2 = *reinterpret_cast<const f32x4*>(F.bcoef + 2 * g.Co + co);
      }
    }
    long n_lo = p_lo / ohw, n_hi = (p_hi - 1) / ohw;
    if constexpr (VOL) { n_lo = n_lo / g.OD * g.SD; n_hi = n_hi / g.OD * g.SD + g.SD - 1; }   // whole volumes of the samples touched
    const long imgs = n_hi - n_lo + 1;
    const unsigned x_bytes = (unsigned)((imgs * g.SH * g.SW - 1) * ld4x + (unsigned)g.SC * 4u);
    rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(x + n_lo * g.SH * g.SW * g.ld_x), 0, (int)x_bytes, 0x00020000);
    x_last = x_bytes - 16u;
    const int co = co0 + ac4 * 4;
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      a_off[i] = co < g.Co ? (unsigned)((tid + 256 * i) / AC4) * ld4y + (unsigned)co * EBY : OOB;
    a_step = BKT * ld4y;
    const int a16 = BKT / g.OW, b16 = BKT - a16 * g.OW;
    c_step = (unsigned)(b16 * g.stride + a16 * g.stride * g.SW) * ld4x;
    c_wrapw = (unsigned)(g.stride * g.SW - g.OW * g.stride) * ld4x;
    c_wraph = (unsigned)(g.SH * g.SW - g.OH * g.stride * g.SW) * ld4x;
    if constexpr (VOL) {      // the next (sample, do) is dstride source images further; past the last do: the next sample's volume
      c_wraph += (unsigned)((g.dstride - 1) * g.SH * g.SW) * ld4x;
      v_wrapd = (unsigned)((g.SD - g.OD * g.dstride) * g.SH * g.SW) * ld4x;
      id_lim = g.OD * g.dstride - g.dpad; v_ods = g.OD * g.dstride; v_dstr = g.dstride;
    }
    dw_step = b16 * g.stride; dh_step = a16 * g.stride;
    ih_lim = g.OH * g.stride - g.pad;
    iw_lim = g.OW * g.stride - g.pad;
    {
      const long p = p_lo + (tid >> 4);
      const long n = p / ohw;
      const int rem = (int)(p - n * ohw);
      const int oh = rem / g.OW, ow = rem - oh * g.OW;
      b_ih = oh * g.stride - g.pad;
      b_iw = ow * g.stride - g.pad;
      long img = n - n_lo;
      if constexpr (VOL) {
        const long smp = n / g.OD;
        const int od = (int)(n - smp * g.OD);
        b_id = od * g.dstride - g.dpad;
        img = smp * g.SD + od * g.dstride - n_lo;
      }
      b_roff = (unsigned)((((int)img * g.SH + oh * g.stride) * g.SW + ow * g.stride) * (long)ld4x);
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int kcol_i = n0 + ((tid & 15) + 16 * i) * 4;
      f_kval[i] = kcol_i < g.Ktot;
      int tap = 0, kc_i = 0;
      if (f_kval[i]) { tap = kcol_i / g.SC; kc_i = kcol_i - tap * g.SC; }
      if constexpr (VOL) { f_kd[i] = tap % g.KD; tap /= g.KD; }
      f_kh[i] = tap / g.KW; f_kw[i] = tap - f_kh[i] * g.KW;
      f_tapc[i] = ((f_kh[i] - g.pad) * g.SW + (f_kw[i] - g.pad)) * (int)ld4x + kc_i * 4;
      if constexpr (VOL) f_tapc[i] += (f_kd[i] - g.dpad) * g.SH * g.SW * (int)ld4x;
      if constexpr (XT == 1) {
        xs[i] = xb[i] = q0;
        if (f_kval[i]) {
          xs[i] = *reinterpret_cast<const f32x4*>(F.xcoef + 2 * g.SC + kc_i);
          xb[i] = *reinterpret_cast<const f32x4*>(F.xcoef + 4 * g.SC + kc_i);
        }
      }
    }
  }
  // the wrap constants live in vector registers (pinned: as scalars each select re-copies its constant every tile)
  int v_ows = g.OW * g.stride, v_ohs = g.OH * g.stride, v_str = g.stride;
  unsigned v_wrapw = c_wrapw, v_wraph = c_wraph;
  if constexpr (FASTLD && !(OCC >= 4 && DYT != 0))      // (the 4-per-CU fused variant has no registers to spare)
    asm volatile("" : "+v"(v_ows), "+v"(v_ohs), "+v"(v_str), "+v"(v_wrapw), "+v"(v_wraph));
  auto load_tile_fast = [&]() {
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      if constexpr (DY16) {
        const u32x2 r = __builtin_bit_cast(u32x2, __builtin_amdgcn_raw_buffer_load_b64(rs_dy, (int)a_off[i], 0, 0));
        const u32x4 wv = {r[0] << 16, r[0] & 0xffff0000u, r[1] << 16, r[1] & 0xffff0000u};
        a_st[i] = __builtin_bit_cast(f32x4, wv);
      } else
        a_st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)a_off[i], 0, 0));
      if constexpr (DYT == 2) {
        a2_st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_dy2, (int)a_off[i], 0, 0));
        a_ok[i] = a_off[i] <= dy_last;       // the hardware range check, restated: rows past the split / channels >= Co
      }
      a_off[i] += a_step;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      bool ok = f_kval[i] && (unsigned)(b_ih + f_kh[i]) < (unsigned)g.SH && (unsigned)(b_iw + f_kw[i]) < (unsigned)g.SW;
      if constexpr (VOL) ok = ok && (unsigned)(b_id + f_kd[VOL ? i : 0]) < (unsigned)g.SD;
      const unsigned off = ok ? b_roff + (unsigned)f_tapc[i] : OOB;
      b_st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
      if constexpr (XT == 1 && MASKX) b_ok[i] = off <= x_last;
    }
    b_iw += dw_step; b_ih += dh_step; b_roff += c_step;
    const bool w = b_iw >= iw_lim;
    b_iw -= w ? v_ows : 0; b_ih += w ? v_str : 0; b_roff += w ? v_wrapw : 0u;
    const bool h = b_ih >= ih_lim;
    b_ih -= h ? v_ohs : 0; b_roff += h ? v_wraph : 0u;
    if constexpr (VOL) {
      b_id += h ? v_dstr : 0;
      const bool d = b_id >= id_lim;
      b_id -= d ? v_ods : 0; b_roff += d ? v_wrapd : 0u;
    }
  };
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto load_tile_vec = [&]() {   // loads the tile the running state points at, then advances the state by one tile
    if constexpr (FASTLD) { load_tile_fast(); return; }
    const int co = co0 + ac4 * 4;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const bool ok = ap[i] < g.P && co < g.Co;
      const f32x4 v = *reinterpret_cast<const f32x4*>(dy + (ok ? ap[i] * g.ld_dy + co : 0));
      a_st[i] = ok ? v : zero4;
      ap[i] += BKT;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int sh = boh[i] * g.stride - g.pad + kkh, sw = bow[i] * g.stride - g.pad + kkw;
      const bool ok = kvalid && bp[i] < g.P && sh >= 0 && sh < g.SH && sw >= 0 && sw < g.SW;
      const long off = ok ? (((long)bn_[i] * g.SH + sh) * g.SW + sw) * g.ld_x + kc : 0;
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + off);
      b_st[i] = ok ? v : zero4;
      bp[i] += BKT; bow[i] += BKT;
      while (bow[i] >= g.OW) { bow[i] -= g.OW; if (++boh[i] == g.OH) { boh[i] = 0; ++bn_[i]; } }
    }
  };
  auto load_tile = [&](int t) {
    if (VEC) { load_tile_vec(); return; }
    const long p0 = (long)t * BKT;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int row = (tid + 256 * i) / AC4;
      const long p = p0 + row;
      const int co = co0 + ac4 * 4;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < g.P) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (co + e < g.Co) v[e] = dy[p * g.ld_dy + co + e];
      }
      a_st[i] = v;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int row = (tid + 256 * i) / BC4;
      const long p = p0 + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < g.P) {
        const int n = (int)(p / ohw);
        const int rem = (int)(p - (long)n * ohw);
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int ke = kcol + e;
          if (ke < g.Ktot) {
            const int tap = ke / g.SC, c = ke - tap * g.SC;
            const int kh = tap / g.KW, kw = tap - kh * g.KW;
            const int sh = oh * g.stride - g.pad + kh, sw = ow * g.stride - g.pad + kw;
            if (sh >= 0 && sh < g.SH && sw >= 0 && sw < g.SW) {
              const long pix = ((long)n * g.SH + sh) * g.SW + sw;
              v[e] = x[pix * g.ld_x + c];
            }
          }
        }
      }
      b_st[i] = v;
    }
  };
  auto store_tile = [&](int buf) {
    if constexpr (SPLW) {
      char* a = (char*)smem + buf * (3 * PLA);
      char* b = (char*)smem + 2 * (3 * PLA) + buf * (3 * PLB);
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        sp_u32x2 p0, p1, p2;
        edrl_split3(a_st[i], p0, p1, p2);
        char* d = a + edrl_wsplit_off<BM>((tid + 256 * i) / AC4, ac4 * 4);
        *reinterpret_cast<sp_u32x2*>(d) = p0;
        *reinterpret_cast<sp_u32x2*>(d + PLA) = p1;
        *reinterpret_cast<sp_u32x2*>(d + 2 * PLA) = p2;
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        sp_u32x2 p0, p1, p2;
        edrl_split3(b_st[i], p0, p1, p2);
        char* d = b + edrl_wsplit_off<BN>(tid >> 4, ((tid & 15) + 16 * i) * 4);
        *reinterpret_cast<sp_u32x2*>(d) = p0;
        *reinterpret_cast<sp_u32x2*>(d + PLB) = p1;
        *reinterpret_cast<sp_u32x2*>(d + 2 * PLB) = p2;
      }
      return;
    }
    float* a = As + buf * BKT * LDA;
    float* b = Bs + buf * BKT * LDB;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const int row = (tid + 256 * i) / AC4;
      *reinterpret_cast<f32x4*>(a + row * LDA + ac4 * 4) = a_st[i];
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      if constexpr (FASTLD) {
        *reinterpret_cast<f32x4*>(b + (tid >> 4) * LDB + ((tid & 15) + 16 * i) * 4) = b_st[i];
      } else {
        const int row = (tid + 256 * i) / BC4;
        *reinterpret_cast<f32x4*>(b + row * LDB + bc4 * 4) = b_st[i];
      }
    }
  };
  // Operand transforms on the staged registers, one element at a time: in the K loop they are issued behind the MFMAs of the
  // tile's second half (matrix-pipe shadow) instead of between the end of the chain and the workgroup barrier.
  constexpr int EL_A = DYT == 2 ? A_LD : 0, EL_B = XT == 1 ? B_LD : 0, EL = EL_A + EL_B;     // 4-channel pieces
  const f32x4 zero4w = {0.f, 0.f, 0.f, 0.f};
  auto transform_piece = [&](int el) {          // el compile-time after unrolling
    if (el < EL_A) {
      if constexpr (DYT == 2) {
        const f32x4 v = edrl_bn_bwd_dx2(a_st[el], a2_st[el], q0, q1, q2);
        a_st[el] = a_ok[el] ? v : zero4w;
      }
    } else if (el < EL) {
      if constexpr (XT == 1) {
        const int i = el - EL_A;
        const f32x4 v = edrl_bn_relu2(b_st[i], xs[i], xb[i]);
        if constexpr (MASKX) b_st[i] = b_ok[i] ? v : zero4w; else b_st[i] = v;
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  if (t_begin < t_end) {
    load_tile(t_begin);
#pragma unroll
    for (int el = 0; el < EL; ++el) transform_piece(el);
    store_tile(0);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
      const int buf = (t - t_begin) & 1;
      if (t + 1 < t_end) load_tile(t + 1);
      if constexpr (SPLW) {
        __builtin_amdgcn_sched_barrier(0);
        const char* pa = (const char*)smem + buf * (3 * PLA);
        const char* pb = (const char*)smem + 2 * (3 * PLA) + buf * (3 * PLB);
        const int cg = 16 * ((lane >> 4) & 1);
        sp_bf16x8 fa[3][TM], fb[2][TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = edrl_wsplit_frag<BN>(pb, 8 * lh, wn0 + 32 * j + cg, lane);
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
          for (int i = 0; i < TM; ++i) fa[p][i] = edrl_wsplit_frag<BM>(pa + p * PLA, 8 * lh, wm0 + 32 * i + cg, lane);
#pragma unroll
        for (int q = 0; q < 3; ++q) {       // im2col plane q meets gradient planes 0 .. 2 - q
          const int cur = q & 1, nxt = cur ^ 1;
          if (q + 1 < 3) {
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[nxt][j] = edrl_wsplit_frag<BN>(pb + (q + 1) * PLB, 8 * lh, wn0 + 32 * j + cg, lane);
          }
#pragma unroll
          for (int p = 2 - q; p >= 0; --p)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[p][i], fb[cur][j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < t_end) {
#pragma unroll
          for (int el = 0; el < EL; ++el) transform_piece(el);
          store_tile(buf ^ 1);
        }
        __syncthreads();
        continue;
      }
      const float* a = As + buf * BKT * LDA + wm0 + li;
      const float* b = Bs + buf * BKT * LDB + wn0 + li;
      // software pipeline over the k-steps: the fragments of step s+1 are requested BEFORE the TM*TN MFMAs of step s
      // are issued (pinned with scheduling-group barriers: left alone the compiler issues them after, and the wave
      // then waits out the LDS latency with an idle MFMA pipe at every step)
      // (volatile: each fragment stays ONE ds_read_b32 with an immediate offset.  Left to merge them, the compiler forms
      // ds_read2 pairs whose 8-bit offsets need a fresh base register per k-step -- 16 VALU adds per tile, and next to the fp32
      // MFMA a VALU instruction is matrix-pipe time while an LDS instruction is not.)
      typedef const volatile __attribute__((address_space(3))) float* lds_vptr;
      const lds_vptr av = (lds_vptr)a;
      const lds_vptr bv = (lds_vptr)b;
      float af[2][TM], bf[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[0][i] = av[lh * LDA + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[0][j] = bv[lh * LDB + j * 32];
      __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);     // step-0 fragments
#pragma unroll
      for (int sidx = 0; sidx < BKT / 2; ++sidx) {
        const int cur = sidx & 1, nxt = cur ^ 1;
        if (sidx + 1 < BKT / 2) {
          const int k = 2 * (sidx + 1) + lh;
#pragma unroll
          for (int i = 0; i < TM; ++i) af[nxt][i] = av[k * LDA + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[nxt][j] = bv[k * LDB + j * 32];
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
        // one 4-channel piece of the NEXT tile's operand transform rides behind the MFMAs of each of the last EL k-steps
        // (its VALU instructions are scheduled as one more group of the step's pipeline)
        const bool xf = EL > 0 && sidx >= BKT / 2 - EL;
        if (xf) transform_piece(sidx - (BKT / 2 - EL));   // (unconditional: on the last tile it re-transforms stale registers that are never stored; a branch here would split the MFMA chain's scheduling region)
        if (sidx + 1 < BKT / 2) __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);   // DS reads
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);                                             // MFMAs
        if (xf) __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);                                          // transform VALU
      }
      if (t + 1 < t_end) store_tile(buf ^ 1);
      __syncthreads();
    }
  }

  float* out = part + (long)split * g.Co * g.Ktot;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn0 + j * 32 + li;
    if (n >= g.Ktot) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (co < g.Co) out[(long)co * g.Ktot + n] = acc[i][j][r];
      }
  }
}

// dW[i] = (accumulate ? dW[i] : 0) + sum_s part[s][i]   (fixed order: deterministic)
// float4 per lane, 8 slabs in flight per iteration (the slab count reaches several hundred for the small layers).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long n,
                                                            int splits, int accumulate) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  if (i4 + 3 < n && (n & 3) == 0) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 7 < splits; z += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(part + (long)(z + u) * n + i4);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; z < splits; ++z) s += *reinterpret_cast<const f32x4*>(part + (long)z * n + i4);
    if (accumulate) s += *reinterpret_cast<const f32x4*>(dw + i4);
    *reinterpret_cast<f32x4*>(dw + i4) = s;
  } else {
    for (long i = i4; i < n && i < i4 + 4; ++i) {
      float s = 0.f;
      for (int z = 0; z < splits; ++z) s += part[(long)z * n + i];
      dw[i] = accumulate ? dw[i] + s : s;
    }
  }
}

#define WG_BK 16
#define WG_OCC 4
#define WG_OCC_FUSED 3
template <int BM, int BN, bool VEC, bool FASTLD = false, int DYT = 0, int XT = 0, bool MASKX = true, bool DY16 = false,
          bool VOL = false>
static int launch_wgrad(const float* dy, const float* x, float* part, const WgradGeom& g, int splits,
                        hipStream_t st, const WgradFuse* fuse = nullptr) {
  constexpr bool SPLW = EDRL_F32_SPLIT != 0 && FASTLD;
  const size_t lds = SPLW ? (size_t)2 * 3 * 16 * (BM + BN) * 2 : (size_t)2 * WG_BK * ((BM + 4) + (BN + 4)) * sizeof(float);
  auto kern = conv_wgrad_f32_kernel<BM, BN, VEC, WG_BK, (XT ? WG_OCC_FUSED : WG_OCC), FASTLD, DYT, XT, MASKX, DY16, VOL>;
  WgradFuse F;
  if (fuse) F = *fuse; else memset(&F, 0, sizeof(F));
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  WgradGeom gg = g;
  gg.tiles_x = edrl_cdiv(g.Ktot, BN); gg.tiles_y = edrl_cdiv(g.Co, BM);
  const long nblk = (long)gg.tiles_x * gg.tiles_y * splits;
  if (nblk > 0x7fffffffL) return EDRL_EINVAL;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, dy, x, part, gg, F);
  EDRL_LAUNCH_CHECK();
  return 0;
}

static void wgrad_plan(long P, int Co, int Ktot, int taps, int* bm, int* bn, int* splits, int* tiles_per_split) {
  *bm = Co <= 64 ? 64 : 128;
  *bn = Ktot <= 64 ? 64 : 128;
  const long tiles = (long)edrl_cdiv(Co, *bm) * edrl_cdiv(Ktot, *bn);
  const long ptiles = (P + WG_BK - 1) / WG_BK;
  // Workgroups per launch (1024 are resident: 256 CUs x 4).  Measured on the ResNet-50 shapes (profiles/): multi-tap
  // convs gain 5-15 % from 3 rounds of shorter pixel ranges (the taps' re-reads of X stay in L2), 1x1 convs are best
  // at 1.5 rounds (fewer partial slabs to reduce).
  const long target_env = edrl_cfg().wgrad_target;
  const long target = target_env > 0 ? target_env : (taps > 1 ? 3072L : 1536L);
  long want = target / tiles;                       // floor: just under a whole number of rounds, never just over
  if (want < 1) want = 1;
  long max_by_len = ptiles / 16; if (max_by_len < 1) max_by_len = 1;  // >= 16 K tiles per split
  long s = want < max_by_len ? want : max_by_len;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  long tps = (ptiles + s - 1) / s;
  s = (ptiles + tps - 1) / tps;
  if (s < 1) s = 1;
  *splits = (int)s;
  *tiles_per_split = (int)tps;
}

// [R][C] -> [C][R] per batch slice (weights [Co][taps][Ci] -> [Ci][taps][Co] uses R=Co, inner handled by caller)
__global__ void permute_021_kernel(const float* __restrict__ in, float* __restrict__ out, int A, int B, int C) {
  // in [A][B][C] -> out [C][B][A]
  const long n = (long)A * B * C;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = (int)(i % A);
  const long t = i / A;
  const int b = (int)(t % B);
  const int c = (int)(t / B);
  out[i] = in[((long)a * B + b) * C + c];
}

extern "C" {

// Kernels of the fp32 implicit-GEMM gather family launched by this process so far (diagnostic: bench.py's per-launch averages).
long edrl_gather_launch_count(void) { return g_gather_launches.load(); }
// 1: this build forms fp32 products as exact bf16x3 splits on the bf16 MFMA (EDRL_F32_SPLIT); 0: fp32 MFMA (libedrl_hip_f32mfma.so)
int edrl_f32_contraction_split(void) { return EDRL_F32_SPLIT != 0 ? 1 : 0; }
// K-split slab of the fp32 gather family: caller-owned, registered per (current device, stream); see gather_slab_locked above.
size_t edrl_gather_ksplit_workspace_bytes(void) { return GATHER_SLAB_BYTES; }
int edrl_gather_ksplit_set_workspace(float* slab, size_t bytes, hipStream_t st) {
  if (slab && bytes < GATHER_SLAB_BYTES) return EDRL_ENOSPC;
  if (slab && ((uintptr_t)slab & 15)) return EDRL_EINVAL;
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
  std::lock_guard<std::mutex> lock(g_slab_mutex);
  for (int i = 0; i < g_slab_n; ++i)
    if (g_slab[i].dev == dev && g_slab[i].stream == st) {
      if (slab) { g_slab[i].ptr = slab; return 0; }
      g_slab[i] = g_slab[--g_slab_n];                   // NULL: forget this (device, stream)
      return 0;
    }
  if (!slab) return 0;
  if (g_slab_n >= GATHER_SLAB_MAX) return EDRL_ENOSPC;
  g_slab[g_slab_n].dev = dev; g_slab[g_slab_n].stream = st; g_slab[g_slab_n].ptr = slab;
  ++g_slab_n;
  return 0;
}

// Convolution forward on NHWC fp32 (also any Linear: KH=KW=1, H=W=1, N=rows).
// y[n,ho,wo,co] = act( sum x[n,ho*s-p+kh,wo*s-p+kw,ci] * w[co,kh,kw,ci] + bias[co] ) * mul + (accum ? y : 0)
int edrl_conv2d_nhwc_fwd_f32(const float* x, const float* w, const float* bias, const float* mul,
                             float* y, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                             int KW, int stride, int pad, long ld_x, long ld_y, long ld_aux, int flags,
                             hipStream_t st) {
  if (N < 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || KH <= 0 || KW <= 0 ||
      stride <= 0 || pad < 0 || ld_x < Ci || ld_y < Co)
    return EDRL_EINVAL;
  if ((long)(Ho - 1) * stride - pad + KH - 1 > (long)Hi - 1 + pad) return EDRL_EINVAL;
  GatherGeom g;
  g.M = (int)((long)N * Ho * Wo);
  if ((long)N * Ho * Wo > 0x7fffffffL) return EDRL_EINVAL;
  g.OH = Ho; g.OW = Wo; g.NC = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Ci;
  g.ld_src = ld_x; g.ld_dst = ld_y; g.ld_aux = ld_aux; g.flags = flags;
  g.h0 = g.w0 = 0; g.step = 1; g.OHs = Ho; g.OWs = Wo;
  g.kh0 = g.kw0 = 0; g.kstep = 1; g.KHs = KH; g.KWs = KW; g.Kfull = g.Ktot; g.sshift = 0;
  g.stat_part = nullptr; g.stat_shift = nullptr;
  return dispatch_gather<false>(x, w, y, bias, mul, g, st);
}

// 3-D convolution forward over NDHWC volumes WITHOUT the depth-unfolded copy (shapes of the reference's
// 3-D networks: baseline_models.py:154-178): the implicit-GEMM gather decodes a third tap level (conv_gather_f32_v2_kernel<..., VOL>).
// x [N,Di,Hi,Wi,Ci], w [Co,KH,KW,KD*Ci] (depth tap innermost of the taps: the weight layout of the depth-unfolded form),
// y [N,Do,Ho,Wo,Co].  Needs Ci % 16 == 0 (vector fast path), Co % 4 == 0, 16-byte aligned tensors; edrl_conv3d_fwd_ok_f32 tells.
int edrl_conv3d_fwd_ok_f32(int N, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co, int KD, int KH, int KW) {
  if (N <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || KD <= 0 || KH <= 0 || KW <= 0) return 0;
  if ((Ci % 16) || (Co % 4) || Ci <= 0 || Co <= 0) return 0;
  const long ovol = (long)Do * Ho * Wo;
  if ((long)N * ovol > 0x7fffffffL) return 0;
  const long ktot = (long)KD * KH * KW * Ci;
  if ((128 / ovol + 2) * (long)Di * Hi * Wi * Ci * 4 >= (1L << 31) || (long)Co * ktot * 4 >= (1L << 31)) return 0;
  return 1;
}
int edrl_conv3d_ndhwc_fwd_f32(const float* x, const float* w, float* y, int N, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo,
                              int Co, int KD, int KH, int KW, int dstride, int stride, int dpad, int pad, hipStream_t st) {
  if (!x || !w || !y || dstride <= 0 || stride <= 0 || dpad < 0 || pad < 0) return EDRL_EINVAL;
  if (!edrl_conv3d_fwd_ok_f32(N, Di, Hi, Wi, Ci, Do, Ho, Wo, Co, KD, KH, KW)) return EDRL_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return EDRL_EINVAL;
  if ((long)(Ho - 1) * stride - pad + KH - 1 > (long)Hi - 1 + pad || (long)(Do - 1) * dstride - dpad + KD - 1 > (long)Di - 1 + dpad)
    return EDRL_EINVAL;
  GatherGeom g;
  g.M = (int)((long)N * Do * Ho * Wo);
  g.OH = Ho; g.OW = Wo; g.NC = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KD * KH * KW * Ci;
  g.ld_src = Ci; g.ld_dst = Co; g.ld_aux = 0; g.flags = GF_VEC_EPI;
  g.h0 = g.w0 = 0; g.step = 1; g.OHs = Ho; g.OWs = Wo;
  g.kh0 = g.kw0 = 0; g.kstep = 1; g.KHs = KH; g.KWs = KW; g.Kfull = g.Ktot; g.sshift = 0;
  g.stat_part = nullptr; g.stat_shift = nullptr;
  g.KD = KD; g.SD = Di; g.OD = Do; g.dstride = dstride; g.dpad = dpad;
  const bool narrow = Co <= 64 || ((long)edrl_cdiv(g.M, 128) * edrl_cdiv(Co, 128) < edrl_cfg().narrow_below);
  if (narrow) return launch_gather_v2<128, 64, false, 16, 4, true, true, 0, 0, true, false, true>(x, w, y, nullptr, nullptr, g, st);
  return launch_gather_v2<128, 128, false, 16, 4, true, true, 0, 0, true, false, true>(x, w, y, nullptr, nullptr, g, st);
}

// Convolution forward with the BatchNorm statistics of its output fused into the epilogue.
// stat_part: [chunks][3][Co] floats with chunks = ceil(N*Ho*Wo / 128) (edrl_conv_stats_chunks); stat_shift is unused
// (reserved: the shift is taken from the data).
// Requires the vector fast path (Ci % 16 == 0, Co % 4 == 0, 16-B aligned tensors): -22 otherwise.
long edrl_conv_stats_chunks(int N, int Ho, int Wo) { return ((long)N * Ho * Wo + 127) / 128; }
int edrl_conv2d_nhwc_fwd_stats_f32(const float* x, const float* w, float* y, const float* stat_shift, float* stat_part,
                                   size_t stat_part_bytes, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co,
                                   int KH, int KW, int stride, int pad, hipStream_t st) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || KH <= 0 || KW <= 0 || stride <= 0 ||
      pad < 0 || (Ci % 16) || (Co % 4) || !stat_part)
    return EDRL_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return EDRL_EINVAL;
  if ((long)N * Ho * Wo > 0x7fffffffL) return EDRL_EINVAL;
  if (stat_part_bytes < (size_t)edrl_conv_stats_chunks(N, Ho, Wo) * 3 * Co * sizeof(float)) return EDRL_ENOSPC;
  GatherGeom g;
  g.M = (int)((long)N * Ho * Wo);
  g.OH = Ho; g.OW = Wo; g.NC = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Ci;
  g.ld_src = Ci; g.ld_dst = Co; g.ld_aux = 0; g.flags = GF_STATS;
  g.h0 = g.w0 = 0; g.step = 1; g.OHs = Ho; g.OWs = Wo;
  g.kh0 = g.kw0 = 0; g.kstep = 1; g.KHs = KH; g.KWs = KW; g.Kfull = g.Ktot; g.sshift = 0;
  g.stat_part = stat_part; g.stat_shift = stat_shift;
  return dispatch_gather<false>(x, w, y, nullptr, nullptr, g, st);
}

// The same contraction (fp32 operands, fp32 MFMA, fp32 BatchNorm partials from the accumulators) with the result stored as bf16:
// the stem of the bf16 trunk (fp32 image in, bf16 raw tensor out -- half the bytes for the passes that follow).  Ho / Wo are the
// caller's (asymmetric bottom / right padding of the space-to-depth stem: ops.stem_conv_fwd).  Ci % 4 == 0, Co % 4 == 0.
int edrl_conv2d_nhwc_fwd_stats_f32_obf16(const float* x, const float* w, void* y_bf16, float* stat_part, size_t stat_part_bytes,
                                         int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride,
                                         int pad, hipStream_t st) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || KH <= 0 || KW <= 0 || stride <= 0 ||
      pad < 0 || (Ci % 4) || (Co % 4) || !stat_part)
    return EDRL_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y_bf16 & 7)) return EDRL_EINVAL;
  if ((long)N * Ho * Wo > 0x7fffffffL) return EDRL_EINVAL;
  if ((long)(Ho - 1) * stride - pad + KH - 1 > (long)Hi - 1 + pad) return EDRL_EINVAL;
  if (stat_part_bytes < (size_t)edrl_conv_stats_chunks(N, Ho, Wo) * 3 * Co * sizeof(float)) return EDRL_ENOSPC;
  GatherGeom g;
  g.M = (int)((long)N * Ho * Wo);
  g.OH = Ho; g.OW = Wo; g.NC = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Ci;
  g.ld_src = Ci; g.ld_dst = Co; g.ld_aux = 0; g.flags = GF_STATS | GF_VEC_EPI;
  g.h0 = g.w0 = 0; g.step = 1; g.OHs = Ho; g.OWs = Wo;
  g.kh0 = g.kw0 = 0; g.kstep = 1; g.KHs = KH; g.KWs = KW; g.Kfull = g.Ktot; g.sshift = 0;
  g.stat_part = stat_part; g.stat_shift = nullptr;
  {   // the space-to-depth stem (16 channels per tap) takes the same buffer-descriptor kernel as its fp32-output twin
      // (dispatch_gather): same tiles, same K loop, so the two outputs differ by the final rounding only
    const long ohw = (long)Ho * Wo;
    const bool fast = (Ci % 16 == 0) && edrl_cfg().gather_buf != 0 && edrl_cfg().gather_variant != 0 && edrl_cfg().gather_variant != 3 &&
                      (128 / ohw + 2) * Hi * Wi * Ci * 4 < (1L << 31) && (long)Co * g.Kfull * 4 < (1L << 31);
    if (fast) {
      if (Co <= 64)
        return launch_gather_v2<128, 64, false, 16, 4, true, true, 0, 0, true, true>(x, w, (float*)y_bf16, nullptr, nullptr, g, st);
      return launch_gather_v2<128, 128, false, 16, 4, true, true, 0, 0, true, true>(x, w, (float*)y_bf16, nullptr, nullptr, g, st);
    }
  }
  if (Co <= 64)
    return launch_gather_v2<128, 64, false, 16, 3, false, false, 0, 0, true, true>(x, w, (float*)y_bf16, nullptr, nullptr, g, st);
  return launch_gather_v2<128, 128, false, 16, 3, false, false, 0, 0, true, true>(x, w, (float*)y_bf16, nullptr, nullptr, g, st);
}

// Convolution data gradient: dx[n,hi,wi,ci] (+)= sum dy[n,ho,wo,co] * wt[ci,kh,kw,co]
// `wt` is the [Ci][KH][KW][Co] permutation of the forward weight (edrl_permute_weight_f32).
int edrl_conv2d_nhwc_dgrad_f32(const float* dy, const float* wt, float* dx, int N, int Hi, int Wi, int Ci,
                               int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, long ld_dy,
                               long ld_dx, int flags, hipStream_t st) {
  if (N < 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || stride <= 0 || pad < 0 ||
      ld_dy < Co || ld_dx < Ci)
    return EDRL_EINVAL;
  if ((long)N * Hi * Wi > 0x7fffffffL) return EDRL_EINVAL;
  int sshift = 0;
  while ((1 << sshift) < stride) ++sshift;
  if ((1 << sshift) != stride) return EDRL_EINVAL;   // power-of-two strides only (ResNet: 1, 2)
  GatherGeom g;
  g.OH = Hi; g.OW = Wi; g.NC = Ci; g.SH = Ho; g.SW = Wo; g.SC = Co;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Kfull = KH * KW * Co;
  g.ld_src = ld_dy; g.ld_dst = ld_dx; g.ld_aux = 0; g.flags = flags;
  g.step = stride; g.kstep = stride; g.sshift = sshift;
  g.stat_part = nullptr; g.stat_shift = nullptr;
  // one launch per parity class (ph, pw) = ((hi+pad) % s, (wi+pad) % s): only the taps kh = ph (mod s) reach it
  for (int ph = 0; ph < stride; ++ph)
    for (int pw = 0; pw < stride; ++pw) {
      g.h0 = ((ph - pad) % stride + stride) % stride;
      g.w0 = ((pw - pad) % stride + stride) % stride;
      g.OHs = g.h0 < Hi ? (Hi - g.h0 + stride - 1) / stride : 0;
      g.OWs = g.w0 < Wi ? (Wi - g.w0 + stride - 1) / stride : 0;
      if (g.OHs == 0 || g.OWs == 0) continue;
      g.kh0 = ph; g.kw0 = pw;
      g.KHs = ph < KH ? (KH - ph + stride - 1) / stride : 0;
      g.KWs = pw < KW ? (KW - pw + stride - 1) / stride : 0;
      g.Ktot = g.KHs * g.KWs * Co;
      if (g.Ktot == 0 && (flags & GF_ACCUM)) continue;   // nothing reaches this class and dx keeps its value
      g.M = (int)((long)N * g.OHs * g.OWs);
      const int rc = dispatch_gather<true>(dy, wt, dx, nullptr, nullptr, g, st);
      if (rc) return rc;
    }
  return 0;
}

// 3-D convolution data gradient over NDHWC volumes without the depth-unfolded gradient and its fold pass (SURVEY.md section 8f
// row 4): dx[n,d,h,w,ci] = sum dy[n,(d+dpad-kd)/ds,(h+pad-kh)/s,(w+pad-kw)/s,co] * w[co,kh,kw,kd,ci] over the taps that divide.
// Depth parity classes (d % dstride) are separate launches like the (h, w) classes of the 2-D data gradient; inside a class the
// source depth is dd + e0 - t for tap kd = q + t*dstride, which is the forward kernel's "rd + tap" with the taps REVERSED -- so
// `wt3` holds, per tap class q = kd % dstride, the matrix [Ci][KH][KW][KDs_q reversed][Co] (edrl_conv3d_dgrad_weight_f32: classes
// concatenated, KD*KH*KW*Ci*Co floats in all) and the kernel runs with depth stride 1.  The destination images of a class are
// dpar + dd*dstride: with Di % dstride == 0 that is "image (n*Dc + dd) of a dstride-times taller tensor", i.e. the strided
// epilogue's n*OH term with OH = Hi*dstride and the base moved by dpar images (needs dstride == 1 or dstride == stride).
int edrl_conv3d_dgrad_ok_f32(int N, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co, int KD, int KH, int KW,
                             int dstride, int stride) {
  if (N <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || KD <= 0 || KH <= 0 || KW <= 0) return 0;
  if ((Co % 16) || (Ci % 4) || Ci <= 0 || Co <= 0 || dstride <= 0 || stride <= 0) return 0;
  if (!(dstride == 1 || dstride == stride) || (stride & (stride - 1)) || (Di % dstride)) return 0;
  if ((long)N * Di * Hi * Wi > 0x7fffffffL) return 0;
  const long cvol = (long)(Di / dstride) * ((Hi + stride - 1) / stride) * ((Wi + stride - 1) / stride);   // rows of one sample in a class (upper bound)
  const long ktot = (long)KD * KH * KW * Co;
  if ((128 / (cvol > 0 ? cvol : 1) + 2) * (long)Do * Ho * Wo * Co * 4 >= (1L << 31) || (long)Ci * ktot * 4 >= (1L << 31)) return 0;
  return 1;
}

__global__ void conv3d_dgrad_weight_kernel(const float* __restrict__ w, float* __restrict__ wt, int Co, int T, int KD, int C,
                                           int dstride) {
  // w [Co][T = KH*KW][KD][C] -> per tap class q: wt_q [C][T][KDs_q][Co], entry t' = tap kd = q + (KDs_q - 1 - t') * dstride
  const long n = (long)Co * T * KD * C;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int c = (int)(i % C);
  long r = i / C;
  const int kd = (int)(r % KD); r /= KD;
  const int t = (int)(r % T);
  const int co = (int)(r / T);
  const int q = kd % dstride, tt = kd / dstride;
  long base = 0;                                   // classes 0 .. q-1 come first
  for (int qq = 0; qq < q; ++qq) base += (long)C * T * ((KD - qq + dstride - 1) / dstride) * Co;
  const int kds = (KD - q + dstride - 1) / dstride;
  wt[base + (((long)c * T + t) * kds + (kds - 1 - tt)) * Co + co] = w[i];
}

int edrl_conv3d_dgrad_weight_f32(const float* w, float* wt3, int Co, int KH, int KW, int KD, int Ci, int dstride, hipStream_t st) {
  if (!w || !wt3 || Co <= 0 || KH <= 0 || KW <= 0 || KD <= 0 || Ci <= 0 || dstride <= 0) return EDRL_EINVAL;
  const long n = (long)Co * KH * KW * KD * Ci;
  if (n > 0x7fffffffL) return EDRL_EINVAL;
  hipLaunchKernelGGL(conv3d_dgrad_weight_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w, wt3, Co, KH * KW, KD, Ci, dstride);
  EDRL_LAUNCH_CHECK();
  return 0;
}

int edrl_conv3d_ndhwc_dgrad_f32(const float* dy, const float* wt3, float* dx, int N, int Di, int Hi, int Wi, int Ci, int Do, int Ho,
                                int Wo, int Co, int KD, int KH, int KW, int dstride, int stride, int dpad, int pad, hipStream_t st) {
  if (!dy || !wt3 || !dx || dpad < 0 || pad < 0) return EDRL_EINVAL;
  if (!edrl_conv3d_dgrad_ok_f32(N, Di, Hi, Wi, Ci, Do, Ho, Wo, Co, KD, KH, KW, dstride, stride)) return EDRL_EINVAL;
  if (((uintptr_t)dy & 15) || ((uintptr_t)wt3 & 15) || ((uintptr_t)dx & 15)) return EDRL_EINVAL;
  int sshift = 0;
  while ((1 << sshift) < stride) ++sshift;
  GatherGeom g;
  g.OH = Hi * dstride; g.OW = Wi; g.NC = Ci; g.SH = Ho; g.SW = Wo; g.SC = Co;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad;
  g.ld_src = Co; g.ld_dst = Ci; g.ld_aux = 0; g.flags = GF_VEC_EPI;
  g.step = stride; g.kstep = stride; g.sshift = sshift;
  g.stat_part = nullptr; g.stat_shift = nullptr;
  const int Dc = Di / dstride;
  for (int dpar = 0; dpar < dstride; ++dpar) {
    const int q = (dpar + dpad) % dstride;                       // tap class that reaches these depths
    const int KDs = q < KD ? (KD - q + dstride - 1) / dstride : 0;
    const int e0 = (dpar + dpad - q) / dstride;
    long wbase = 0;
    for (int qq = 0; qq < q; ++qq) wbase += (long)Ci * KH * KW * ((KD - qq + dstride - 1) / dstride) * Co;
    g.KD = KDs; g.SD = Do; g.OD = Dc; g.dstride = 1; g.dpad = KDs - 1 - e0;
    g.Kfull = KH * KW * KDs * Co;
    float* dxc = dx + (long)dpar * Hi * Wi * Ci;
    for (int ph = 0; ph < stride; ++ph)
      for (int pw = 0; pw < stride; ++pw) {
        g.h0 = ((ph - pad) % stride + stride) % stride;
        g.w0 = ((pw - pad) % stride + stride) % stride;
        g.OHs = g.h0 < Hi ? (Hi - g.h0 + stride - 1) / stride : 0;
        g.OWs = g.w0 < Wi ? (Wi - g.w0 + stride - 1) / stride : 0;
        if (g.OHs == 0 || g.OWs == 0) continue;
        g.kh0 = ph; g.kw0 = pw;
        g.KHs = (ph < KH && KDs > 0) ? (KH - ph + stride - 1) / stride : 0;
        g.KWs = pw < KW ? (KW - pw + stride - 1) / stride : 0;
        g.Ktot = g.KHs * g.KWs * KDs * Co;
        g.M = (int)((long)N * Dc * g.OHs * g.OWs);
        const bool narrow = Ci <= 64 || ((long)edrl_cdiv(g.M, 128) * edrl_cdiv(Ci, 128) < edrl_cfg().narrow_below);
        const int rc = narrow ? launch_gather_v2<128, 64, true, 16, 4, true, true, 0, 0, true, false, true>(dy, wt3 + wbase, dxc, nullptr, nullptr, g, st)
                              : launch_gather_v2<128, 128, true, 16, 4, true, true, 0, 0, true, false, true>(dy, wt3 + wbase, dxc, nullptr, nullptr, g, st);
        if (rc) return rc;
      }
  }
  return 0;
}

size_t edrl_conv2d_nhwc_wgrad_workspace_bytes(int N, int Ho, int Wo, int Co, int Ci, int KH, int KW) {
  int bm, bn, splits, tps;
  wgrad_plan((long)N * Ho * Wo, Co, KH * KW * Ci, KH * KW, &bm, &bn, &splits, &tps);
  return (size_t)splits * Co * KH * KW * Ci * sizeof(float);
}

}  // extern "C"

// Convolution weight gradient: dw[co,kh,kw,ci] (+)= sum_pix dy[pix,co] * x[pix @ tap, ci]
static bool wgrad_fast_ok(const float* dy, const float* x, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, long ld_dy, long ld_x,
                          int tiles_per_split) {
  const bool vec = (Ci % 4 == 0) && (ld_x % 4 == 0) && (Co % 4 == 0) && (ld_dy % 4 == 0) &&
                   (((uintptr_t)dy & 15) == 0) && (((uintptr_t)x & 15) == 0);
  // buffer-load path: decode wraps at most once per tile, per-block operand footprints addressable with 31 bits
  const long span = (long)tiles_per_split * WG_BK;
  return vec && (WG_BK / Wo + 1 <= Ho) && span * ld_dy * 4 < (1L << 31) &&
         (span / ((long)Ho * Wo) + 2) * Hi * Wi * ld_x * 4 < (1L << 31);
}

template <int DYT, int XT, bool MASKX>
static int wgrad_fused_launch(const float* dy, const float* x, float* ws, const WgradGeom& g, int bm, int bn, int splits,
                              const WgradFuse& F, hipStream_t st) {
  if (bm == 64 && bn == 64) return launch_wgrad<64, 64, true, true, DYT, XT, MASKX>(dy, x, ws, g, splits, st, &F);
  if (bm == 64) return launch_wgrad<64, 128, true, true, DYT, XT, MASKX>(dy, x, ws, g, splits, st, &F);
  if (bn == 64) return launch_wgrad<128, 64, true, true, DYT, XT, MASKX>(dy, x, ws, g, splits, st, &F);
  return launch_wgrad<128, 128, true, true, DYT, XT, MASKX>(dy, x, ws, g, splits, st, &F);
}

static int wgrad_impl(const float* dy, const float* x, float* dw, float* workspace, size_t workspace_bytes, int N, int Hi,
                      int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, long ld_dy, long ld_x,
                      int accumulate, const WgradFuse* fuse, hipStream_t st) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || stride <= 0 || pad < 0 ||
      ld_dy < Co || ld_x < Ci)
    return EDRL_EINVAL;
  WgradGeom g;
  g.P = (long)N * Ho * Wo;
  if (g.P > 0x7fffffffL) return EDRL_EINVAL;      // (the kernels count pixel tiles in 32 bits)
  g.OH = Ho; g.OW = Wo; g.Co = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Ci;
  g.ld_dy = ld_dy; g.ld_x = ld_x;
  int bm, bn, splits;
  wgrad_plan(g.P, Co, g.Ktot, KH * KW, &bm, &bn, &splits, &g.tiles_per_split);
  const size_t need = (size_t)splits * Co * g.Ktot * sizeof(float);
  if (workspace_bytes < need || workspace == nullptr) return EDRL_ENOSPC;
  const bool vec = (Ci % 4 == 0) && (ld_x % 4 == 0) && (Co % 4 == 0) && (ld_dy % 4 == 0) &&
                   (((uintptr_t)dy & 15) == 0) && (((uintptr_t)x & 15) == 0);
  const bool fast_env = edrl_cfg().wgrad_fast != 0;
  const bool fast_ok = wgrad_fast_ok(dy, x, Hi, Wi, Ci, Ho, Wo, Co, ld_dy, ld_x, g.tiles_per_split);
  const bool fast = fast_ok && fast_env;
  int rc;
  if (fuse) {
    if (!fast_ok) return EDRL_EINVAL;     // the fused operands exist on the buffer-load path only
    const bool dyt = fuse->dy2 != nullptr, xt = fuse->xcoef != nullptr;
    if (dyt && (((uintptr_t)fuse->dy2 & 15) != 0)) return EDRL_EINVAL;
    const bool maskx = !(KH == 1 && KW == 1 && pad == 0);
    if (dyt && xt && maskx) rc = wgrad_fused_launch<2, 1, true>(dy, x, workspace, g, bm, bn, splits, *fuse, st);
    else if (dyt && xt) rc = wgrad_fused_launch<2, 1, false>(dy, x, workspace, g, bm, bn, splits, *fuse, st);
    else if (dyt) rc = wgrad_fused_launch<2, 0, true>(dy, x, workspace, g, bm, bn, splits, *fuse, st);
    else if (xt && maskx) rc = wgrad_fused_launch<0, 1, true>(dy, x, workspace, g, bm, bn, splits, *fuse, st);     // (dY materialised)
    else if (xt) rc = wgrad_fused_launch<0, 1, false>(dy, x, workspace, g, bm, bn, splits, *fuse, st);
    else return EDRL_EINVAL;
  } else if (fast) {
    if (bm == 64 && bn == 64) rc = launch_wgrad<64, 64, true, true>(dy, x, workspace, g, splits, st);
    else if (bm == 64) rc = launch_wgrad<64, 128, true, true>(dy, x, workspace, g, splits, st);
    else if (bn == 64) rc = launch_wgrad<128, 64, true, true>(dy, x, workspace, g, splits, st);
    else rc = launch_wgrad<128, 128, true, true>(dy, x, workspace, g, splits, st);
  } else if (bm == 64 && bn == 64)
    rc = vec ? launch_wgrad<64, 64, true>(dy, x, workspace, g, splits, st)
             : launch_wgrad<64, 64, false>(dy, x, workspace, g, splits, st);
  else if (bm == 64)
    rc = vec ? launch_wgrad<64, 128, true>(dy, x, workspace, g, splits, st)
             : launch_wgrad<64, 128, false>(dy, x, workspace, g, splits, st);
  else if (bn == 64)
    rc = vec ? launch_wgrad<128, 64, true>(dy, x, workspace, g, splits, st)
             : launch_wgrad<128, 64, false>(dy, x, workspace, g, splits, st);
  else
    rc = vec ? launch_wgrad<128, 128, true>(dy, x, workspace, g, splits, st)
             : launch_wgrad<128, 128, false>(dy, x, workspace, g, splits, st);
  if (rc) return rc;
  const long n = (long)Co * g.Ktot;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(edrl_cdiv(n, 1024)), dim3(256), 0, st, workspace, dw, n,
                     splits, accumulate);
  EDRL_LAUNCH_CHECK();
  return 0;
}

extern "C" {
int edrl_conv2d_nhwc_wgrad_f32(const float* dy, const float* x, float* dw, float* workspace,
                               size_t workspace_bytes, int N, int Hi, int Wi, int Ci, int Ho, int Wo,
                               int Co, int KH, int KW, int stride, int pad, long ld_dy, long ld_x,
                               int accumulate, hipStream_t st) {
  return wgrad_impl(dy, x, dw, workspace, workspace_bytes, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, ld_dy, ld_x,
                    accumulate, nullptr, st);
}

// 3-D convolution weight gradient over NDHWC volumes without the depth-unfolded operand:
// dw[co,kh,kw,kd*Ci+ci] = sum dy[n,do,ho,wo,co] * x[n,do*ds-dpad+kd,ho*s-pad+kh,wo*s-pad+kw,ci] -- the weight layout of the
// depth-unfolded form.  Workspace: edrl_conv2d_nhwc_wgrad_workspace_bytes(N*Do, Ho, Wo, Co, KD*Ci, KH, KW) (the same split-K plan).
// Ci % 4 == 0, Co % 4 == 0, 16-byte aligned dense tensors, buffer-load path geometry: edrl_conv3d_wgrad_ok_f32.
int edrl_conv3d_wgrad_ok_f32(int N, int Di, int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co, int KD, int KH, int KW) {
  if (N <= 0 || Di <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || KD <= 0 || KH <= 0 || KW <= 0)
    return 0;
  if ((Ci % 4) || (Co % 4)) return 0;
  const long P = (long)N * Do * Ho * Wo;
  if (P > 0x7fffffffL || (long)KD * KH * KW * Ci > 0x7fffffffL) return 0;
  int bm, bn, splits, tps;
  wgrad_plan(P, Co, KD * KH * KW * Ci, KD * KH * KW, &bm, &bn, &splits, &tps);
  const long span = (long)tps * WG_BK;
  return (WG_BK / Wo + 1 <= Ho) && span * Co * 4 < (1L << 31) &&
         (span / ((long)Do * Ho * Wo) + 2) * Di * Hi * Wi * Ci * 4 < (1L << 31);
}
// Workspace of edrl_conv3d_ndhwc_wgrad_f32: the launcher's own split-K plan (tap count KD*KH*KW; the 2-D helper called with
// KD*Ci channels plans with KH*KW taps and can come out smaller for KD > 1 with KH = KW = 1).
size_t edrl_conv3d_wgrad_workspace_bytes(int N, int Do, int Ho, int Wo, int Co, int Ci, int KD, int KH, int KW) {
  if (N <= 0 || Do <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || Ci <= 0 || KD <= 0 || KH <= 0 || KW <= 0) return 0;
  int bm, bn, splits, tps;
  const long K = (long)KD * KH * KW * Ci;
  wgrad_plan((long)N * Do * Ho * Wo, Co, (int)K, KD * KH * KW, &bm, &bn, &splits, &tps);
  return (size_t)splits * Co * K * sizeof(float);
}
int edrl_conv3d_ndhwc_wgrad_f32(const float* dy, const float* x, float* dw, float* workspace, size_t workspace_bytes, int N, int Di,
                                int Hi, int Wi, int Ci, int Do, int Ho, int Wo, int Co, int KD, int KH, int KW, int dstride,
                                int stride, int dpad, int pad, int accumulate, hipStream_t st) {
  if (!dy || !x || !dw || dstride <= 0 || stride <= 0 || dpad < 0 || pad < 0) return EDRL_EINVAL;
  if (!edrl_conv3d_wgrad_ok_f32(N, Di, Hi, Wi, Ci, Do, Ho, Wo, Co, KD, KH, KW)) return EDRL_EINVAL;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15) || ((uintptr_t)dw & 15)) return EDRL_EINVAL;
  if ((long)(Ho - 1) * stride - pad + KH - 1 > (long)Hi - 1 + pad || (long)(Do - 1) * dstride - dpad + KD - 1 > (long)Di - 1 + dpad)
    return EDRL_EINVAL;
  WgradGeom g;
  g.P = (long)N * Do * Ho * Wo;
  g.OH = Ho; g.OW = Wo; g.Co = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KD * KH * KW * Ci;
  g.ld_dy = Co; g.ld_x = Ci;
  g.KD = KD; g.SD = Di; g.OD = Do; g.dstride = dstride; g.dpad = dpad;
  int bm, bn, splits;
  wgrad_plan(g.P, Co, g.Ktot, KD * KH * KW, &bm, &bn, &splits, &g.tiles_per_split);
  if (workspace_bytes < (size_t)splits * Co * g.Ktot * sizeof(float) || workspace == nullptr) return EDRL_ENOSPC;
  int rc;
  if (bm == 64 && bn == 64) rc = launch_wgrad<64, 64, true, true, 0, 0, true, false, true>(dy, x, workspace, g, splits, st);
  else if (bm == 64) rc = launch_wgrad<64, 128, true, true, 0, 0, true, false, true>(dy, x, workspace, g, splits, st);
  else if (bn == 64) rc = launch_wgrad<128, 64, true, true, 0, 0, true, false, true>(dy, x, workspace, g, splits, st);
  else rc = launch_wgrad<128, 128, true, true, 0, 0, true, false, true>(dy, x, workspace, g, splits, st);
  if (rc) return rc;
  const long n = (long)Co * g.Ktot;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(edrl_cdiv(n, 1024)), dim3(256), 0, st, workspace, dw, n, splits, accumulate);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// The same weight gradient with dy stored as bf16 (dense [N,Ho,Wo,Co]) and x fp32: the stem of the bf16 trunk, whose d_raw comes out
// of edrl_maxpool3x3s2_bn_bwd_apply_mx as bf16.  Co <= 64, Co % 4 == 0, Ci % 4 == 0, buffer-load path geometry (-22 otherwise);
// workspace as edrl_conv2d_nhwc_wgrad_workspace_bytes.
// 1 when the geometry is served by edrl_conv2d_nhwc_wgrad_f32_dybf16 (dense tensors), 0 otherwise (widen dy and use the fp32 entry).
int edrl_conv2d_nhwc_wgrad_f32_dybf16_ok(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || Co > 64 || KH <= 0 || KW <= 0) return 0;
  const long P = (long)N * Ho * Wo;
  if (P > 0x7fffffffL) return 0;
  int bm, bn, splits, tps;
  wgrad_plan(P, Co, KH * KW * Ci, KH * KW, &bm, &bn, &splits, &tps);
  return wgrad_fast_ok((const float*)nullptr, (const float*)nullptr, Hi, Wi, Ci, Ho, Wo, Co, Co, Ci, tps) ? 1 : 0;
}
int edrl_conv2d_nhwc_wgrad_f32_dybf16(const void* dy_bf16, const float* x, float* dw, float* workspace, size_t workspace_bytes,
                                      int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad,
                                      int accumulate, hipStream_t st) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || Co > 64 || stride <= 0 || pad < 0 || !dy_bf16)
    return EDRL_EINVAL;
  WgradGeom g;
  g.P = (long)N * Ho * Wo;
  if (g.P > 0x7fffffffL) return EDRL_EINVAL;
  g.OH = Ho; g.OW = Wo; g.Co = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Ci;
  g.ld_dy = Co; g.ld_x = Ci;
  int bm, bn, splits;
  wgrad_plan(g.P, Co, g.Ktot, KH * KW, &bm, &bn, &splits, &g.tiles_per_split);
  if (workspace_bytes < (size_t)splits * Co * g.Ktot * sizeof(float) || workspace == nullptr) return EDRL_ENOSPC;
  if (((uintptr_t)dy_bf16 & 7) || !wgrad_fast_ok((const float*)nullptr, x, Hi, Wi, Ci, Ho, Wo, Co, Co, Ci, g.tiles_per_split))
    return EDRL_EINVAL;
  const int rc = bn == 64 ? launch_wgrad<64, 64, true, true, 0, 0, true, true>((const float*)dy_bf16, x, workspace, g, splits, st)
                          : launch_wgrad<64, 128, true, true, 0, 0, true, true>((const float*)dy_bf16, x, workspace, g, splits, st);
  if (rc) return rc;
  const long n = (long)Co * g.Ktot;
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(edrl_cdiv(n, 1024)), dim3(256), 0, st, workspace, dw, n, splits, accumulate);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// Weight gradient with the BatchNorm passes on either side folded into the operand loads (dense NHWC tensors):
//   dY = A*g - K1 - K2*(yraw - mean)      g = masked gradient of this layer's BatchNorm output, yraw = this layer's raw conv
//                                          output, bcoef [4][Co] = {A, nK2, C2, mean} (edrl_bn_bwd_finalize_partials_f32)
//   X  = relu(x*scale + shift2)            when x_fcoef [5][Ci] = {mean, rstd, scale, shift, shift2} of the PREVIOUS layer's BatchNorm
//                                          is given (x is then that layer's raw conv output); x as is when x_fcoef == NULL
//   yraw == bcoef == NULL (with x_fcoef): g IS dY, a materialised d_raw (encoders._K32 draw_sep units)
// Needs the buffer-load fast path (edrl_conv2d_fused_ok_f32): -22 otherwise.
int edrl_conv2d_nhwc_wgrad_bn_f32(const float* g, const float* yraw, const float* bcoef, const float* x,
                                  const float* x_fcoef, float* dw, float* workspace, size_t workspace_bytes, int N, int Hi,
                                  int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, int accumulate,
                                  hipStream_t st) {
  if (!g || (!yraw != !bcoef) || !x) return EDRL_EINVAL;
  if (!yraw && !x_fcoef) return EDRL_EINVAL;        // (no transform on either side: that is edrl_conv2d_nhwc_wgrad_f32)
  WgradFuse F;
  F.dy2 = yraw; F.bcoef = bcoef; F.xcoef = x_fcoef;
  return wgrad_impl(g, x, dw, workspace, workspace_bytes, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, Co, Ci, accumulate,
                    &F, st);
}

// ---- fused-BatchNorm forward / data gradient (dense NHWC tensors; conv_geom.h GatherFuse)
// Forward conv whose INPUT is the raw conv output of the previous layer: a = relu((x - mean)*scale + shift) with
// in_fcoef [5][Ci] = {mean, rstd, scale, shift, shift2} (edrl_bn_finalize_partials_f32) is formed in the operand load, and the
// BatchNorm chunk partials of the OUTPUT are emitted like edrl_conv2d_nhwc_fwd_stats_f32.
int edrl_conv2d_nhwc_fwd_bnin_stats_f32(const float* x, const float* in_fcoef, const float* w, float* y, float* stat_part,
                                        size_t stat_part_bytes, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                                        int KW, int stride, int pad, hipStream_t st) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || KH <= 0 || KW <= 0 || stride <= 0 ||
      pad < 0 || (Ci % 16) || (Co % 4) || !stat_part || !in_fcoef)
    return EDRL_EINVAL;
  if ((long)N * Ho * Wo > 0x7fffffffL) return EDRL_EINVAL;
  if (stat_part_bytes < (size_t)edrl_conv_stats_chunks(N, Ho, Wo) * 3 * Co * sizeof(float)) return EDRL_ENOSPC;
  GatherGeom g;
  g.M = (int)((long)N * Ho * Wo);
  g.OH = Ho; g.OW = Wo; g.NC = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Ci;
  g.ld_src = Ci; g.ld_dst = Co; g.ld_aux = 0; g.flags = GF_STATS;
  g.h0 = g.w0 = 0; g.step = 1; g.OHs = Ho; g.OWs = Wo;
  g.kh0 = g.kw0 = 0; g.kstep = 1; g.KHs = KH; g.KWs = KW; g.Kfull = g.Ktot; g.sshift = 0;
  g.stat_part = stat_part; g.stat_shift = nullptr;
  GatherFuse F;
  memset(&F, 0, sizeof(F));
  F.acoef = in_fcoef;
  return dispatch_gather_fused<false, 1, 0>(x, w, y, g, F, st);
}

// Chunks (128-row tiles, summed over the parity classes of a strided layer) of the partial sums the fused data gradient emits.
long edrl_conv_dgrad_bn_chunks(int N, int Hi, int Wi, int stride, int pad) {
  long chunks = 0;
  for (int ph = 0; ph < stride; ++ph)
    for (int pw = 0; pw < stride; ++pw) {
      const int h0 = ((ph - pad) % stride + stride) % stride, w0 = ((pw - pad) % stride + stride) % stride;
      const long ohs = h0 < Hi ? (Hi - h0 + stride - 1) / stride : 0, ows = w0 < Wi ? (Wi - w0 + stride - 1) / stride : 0;
      chunks += ((long)N * ohs * ows + 127) / 128;
    }
  return chunks;
}

// Data gradient with the BatchNorm-backward passes on both sides folded in:
//   operand   dY = A*g + nK2*yraw + C2, bcoef [4][Co] = {A, nK2, C2, mean}; yraw == bcoef == NULL: g_in IS dY (a materialised
//             d_raw: units that run on the plain operand path but keep the epilogue -- encoders mid_sep / wide blocks)
//   epilogue  (ep_raw != NULL) dx is the gradient of relu?(bn(ep_raw)) of the layer below: it is masked with ep_mask (sign
//             bytes [pixel][Ci/4]) or, when ep_mask == NULL and ep_relu, with the decision recomputed from ep_raw and
//             ep_fcoef [5][Ci] = {mean, rstd, scale, shift, shift2}; the masked gradient is stored and (sum g, sum g*xhat) per
//             128-row tile go to ep_part [edrl_conv_dgrad_bn_chunks][2][Ci].
//   flags     GF_ACCUM (2): dx += (before masking / reduction).
int edrl_conv2d_nhwc_dgrad_bn_f32(const float* g_in, const float* yraw, const float* bcoef, const float* wt, float* dx, int N,
                                  int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad,
                                  int flags, const float* ep_raw, const unsigned char* ep_mask, const float* ep_fcoef,
                                  int ep_relu, float* ep_part, size_t ep_part_bytes, hipStream_t st) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || stride <= 0 || pad < 0 || !g_in ||
      (!yraw != !bcoef) || (Co % 16) || (Ci % 4))
    return EDRL_EINVAL;
  const bool plain_in = !yraw;
  if (plain_in && !ep_raw) return EDRL_EINVAL;      // (no transform on either side: that is edrl_conv2d_nhwc_dgrad_f32)
  if ((long)N * Hi * Wi > 0x7fffffffL) return EDRL_EINVAL;
  if (ep_raw && (!ep_fcoef || !ep_part)) return EDRL_EINVAL;
  if (ep_raw && ep_part_bytes < (size_t)edrl_conv_dgrad_bn_chunks(N, Hi, Wi, stride, pad) * 2 * Ci * sizeof(float))
    return EDRL_ENOSPC;
  int sshift = 0;
  while ((1 << sshift) < stride) ++sshift;
  if ((1 << sshift) != stride) return EDRL_EINVAL;
  GatherGeom g;
  g.OH = Hi; g.OW = Wi; g.NC = Ci; g.SH = Ho; g.SW = Wo; g.SC = Co;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Kfull = KH * KW * Co;
  g.ld_src = Co; g.ld_dst = Ci; g.ld_aux = 0;
  g.flags = (flags & GF_ACCUM) | ((ep_raw && !ep_mask && ep_relu) ? GF_EPI_RELU : 0);
  g.step = stride; g.kstep = stride; g.sshift = sshift;
  g.stat_part = nullptr; g.stat_shift = nullptr;
  GatherFuse F;
  memset(&F, 0, sizeof(F));
  F.src2 = yraw;
  F.acoef = bcoef;
  if (ep_raw) {
    F.ep_x = ep_raw; F.ld_ep = Ci; F.ep_mask = ep_mask; F.ep_fcoef = ep_fcoef; F.ep_part = ep_part;
  }
  int chunk0 = 0;
  for (int ph = 0; ph < stride; ++ph)
    for (int pw = 0; pw < stride; ++pw) {
      g.h0 = ((ph - pad) % stride + stride) % stride;
      g.w0 = ((pw - pad) % stride + stride) % stride;
      g.OHs = g.h0 < Hi ? (Hi - g.h0 + stride - 1) / stride : 0;
      g.OWs = g.w0 < Wi ? (Wi - g.w0 + stride - 1) / stride : 0;
      if (g.OHs == 0 || g.OWs == 0) continue;
      g.kh0 = ph; g.kw0 = pw;
      g.KHs = ph < KH ? (KH - ph + stride - 1) / stride : 0;
      g.KWs = pw < KW ? (KW - pw + stride - 1) / stride : 0;
      g.Ktot = g.KHs * g.KWs * Co;
      g.M = (int)((long)N * g.OHs * g.OWs);
      // a class no tap reaches: dx keeps its value when accumulating and nothing is reduced; otherwise the kernel still runs
      // (zero fill, or mask + reduce the accumulated gradient)
      if (g.Ktot == 0 && (flags & GF_ACCUM) && !ep_raw) continue;
      F.ep_chunk0 = chunk0;
      // EPI 2 = the instantiation whose lean epilogue is specialised for (sign bytes, accumulate); EPI 1 for (recomputed
      // ReLU decision, no accumulate); both carry the general epilogue for every other combination
      const int rc = plain_in ? (ep_mask ? dispatch_gather_fused<true, 0, 2>(g_in, wt, dx, g, F, st)
                                         : dispatch_gather_fused<true, 0, 1>(g_in, wt, dx, g, F, st))
                     : !ep_raw ? dispatch_gather_fused<true, 2, 0>(g_in, wt, dx, g, F, st)
                     : (ep_mask && (flags & GF_ACCUM)) ? dispatch_gather_fused<true, 2, 2>(g_in, wt, dx, g, F, st)
                                                       : dispatch_gather_fused<true, 2, 1>(g_in, wt, dx, g, F, st);
      if (rc) return rc;
      chunk0 += (g.M + 127) / 128;
    }
  return 0;
}

// 1 when every fused variant (forward operand transform, data gradient, weight gradient) has its fast path for this layer
// geometry with dense, 16-byte-aligned tensors; the host side (encoders.py) falls back to the separate BatchNorm passes if not.
int edrl_conv2d_fused_ok_f32(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || (Ci % 16) || (Co % 16) || (stride != 1 && stride != 2)) return 0;
  if ((long)N * Hi * Wi > 0x7fffffffL || (long)N * Ho * Wo > 0x7fffffffL) return 0;
  const long kfull = (long)KH * KW * Ci;
  // forward gather: source = input, rows = output pixels
  if (!((128 / ((long)Ho * Wo) + 2) * Hi * Wi * Ci * 4 < (1L << 31) && (long)Co * kfull * 4 < (1L << 31))) return 0;
  // data gradient: source = output-gradient, rows = input pixels of one parity class (>= 1 pixel per image)
  const long cls = ((long)(Hi + stride - 1) / stride) * ((Wi + stride - 1) / stride);
  if (!((128 / (cls > 0 ? cls : 1) + 2 + 1) * Ho * Wo * Co * 4 < (1L << 31))) return 0;
  int bm, bn, splits, tps;
  wgrad_plan((long)N * Ho * Wo, Co, (int)kfull, KH * KW, &bm, &bn, &splits, &tps);
  return wgrad_fast_ok((const float*)nullptr, (const float*)nullptr, Hi, Wi, Ci, Ho, Wo, Co, Co, Ci, tps) ? 1 : 0;
}

// in [A][B][C] -> out [C][B][A]   (weight [Co][taps][Ci] -> [Ci][taps][Co]; B=1 gives a matrix transpose)
int edrl_permute_weight_f32(const float* in, float* out, int A, int B, int C, hipStream_t st) {
  if (A <= 0 || B <= 0 || C <= 0) return EDRL_EINVAL;
  const long n = (long)A * B * C;
  hipLaunchKernelGGL(permute_021_kernel, dim3(edrl_cdiv(n, 256)), dim3(256), 0, st, in, out, A, B, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
