// bf16 implicit-GEMM convolution (forward / data gradient) on the gfx950 bf16 matrix cores — the C2/C4 precision
// of SURVEY.md §8(a) rows E1/E2: bf16 operands, fp32 accumulate, fp32 BatchNorm statistics taken from the accumulators.
//
// Same contraction and tiling as conv_gemm.hip's fast path: D[m][n] = sum_k Agather[m][k] * Wm[n][k], M = output
// pixels (NHWC rows), N = Cout, K = (kh,kw,c); 256 threads = 2x2 waves, wave tile 64x64 of 32x32 MFMA tiles
// (v_mfma_f32_32x32x16_bf16: lane (r = l&31, h = l>>5) supplies A[r][8h..8h+7] / B[8h..8h+7][r] as ONE 16-byte
// fragment), K tile 32 bf16 = 64 B per row, staged to LDS as [row][32+8] (80-B rows: ds_read_b128 conflict-free),
// double buffered, 40 KiB per workgroup -> 3 workgroups per CU.  The source channel count must be a multiple of 32
// (a K tile never straddles a tap), which holds for every conv of the ResNet trunks except the stem; the stem and
// the weight gradient stay on the fp32 kernels (conv_gemm.hip) in round 1.
#include "edrl_common.h"
#include "edrl_config.h"
#include <stdlib.h>
#include <string.h>
#include "conv_geom.h"
#include <type_traits>
#include "conv_bf16_v3.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define HBK 32            // K tile in bf16 elements
#define HLD (HBK + 8)     // padded LDS row (elements): 80 bytes

typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));

// BUF: operand loads through buffer descriptors exactly as in conv_gemm.hip's fp32 kernel (masked rows / taps are an
// out-of-range offset the hardware zero-fills; one add per staged piece and tile).
//
// Fused BatchNorm (BUF only; conv_geom.h GatherFuse), the bf16 counterpart of conv_gemm.hip's ATR / EPI variants -- next to the
// bf16 MFMA the VALU is a separate pipe, so the transforms are nearly free and the kernel stays HBM-bound on most layers:
//   ATR 1: the gathered operand is a RAW (bf16) conv output; bf16(relu(x*scale + shift2)) is formed while the tile is staged.
//   ATR 2: the gathered operand is d_raw = bf16(A*g + nK2*x + C2) from the masked gradient g (src) and the raw output x (src2).
//   EPI 1: the computed tile is the gradient of a BatchNorm(+ReLU) output: masked (sign bytes or decision re-derived from the
//          raw tensor), stored as bf16, and (sum g, sum g*(x - mean)) of the UNROUNDED fp32 values go to ep_part.
//   MASK : padding taps / masked rows must read as exactly 0 after the transform (false for 1x1 / pad-0 layers).
// OCC4: compiled for 4 workgroups per CU (128 VGPRs, 4 x 40 KiB of LDS): the forward kernels (plain and with the operand transform,
// ATR 1): 8 MFMAs per wave and K tile cannot hide a global load, so the K loop runs at about one memory latency per tile and a
// fourth workgroup per CU is worth more than the 4 spilled registers (a second register stage for the loads was tried instead:
// 46 spilled registers at the 168-register budget, 2x slower).
template <int BN, bool DGRAD, bool BUF = false, int ATR = 0, int EPI = 0, bool MASK = true, bool OCC4 = false>
__global__ __launch_bounds__(256, OCC4 ? 4 : 3) void conv_gather_bf16_kernel(const __bf16* __restrict__ src,
                                                                  const __bf16* __restrict__ wm,
                                                                  __bf16* __restrict__ dst, GatherGeom g, int tiles_n,
                                                                  GatherFuse F) {
  static_assert(ATR == 0 || BUF, "operand transforms ride on the buffer-descriptor path");
  constexpr int BM = 128;
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int A_LD = BM / 64, B_LD = BN / 64;   // 16-byte loads per thread per K tile (64 rows per pass)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);      // [2][BM][HLD]
  __bf16* Bs = As + 2 * BM * HLD;                        // [2][BN][HLD]
  float* smem = reinterpret_cast<float*>(smem_raw);      // epilogue staging view

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  const int lid = edrl_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const long m0 = (long)tile_m * BM;
  const int n0 = tile_n * BN;

  const int k8 = (tid & 3) * 8;
  const int r0 = tid >> 2;
  int rn[A_LD], rh[A_LD], rw[A_LD];
#pragma unroll
  for (int i = 0; i < A_LD; ++i) {
    const long m = m0 + r0 + 64 * i;
    if (m < g.M) {
      const int ohw = g.OHs * g.OWs;
      const int n = (int)(m / ohw);
      const int rem = (int)(m - (long)n * ohw);
      const int ii = rem / g.OWs, jj = rem - ii * g.OWs;
      const int oh = g.h0 + ii * g.step, ow = g.w0 + jj * g.step;
      rn[i] = n;
      if (DGRAD) { rh[i] = oh + g.pad; rw[i] = ow + g.pad; }
      else       { rh[i] = oh * g.stride - g.pad; rw[i] = ow * g.stride - g.pad; }
    } else { rn[i] = -1; rh[i] = 0; rw[i] = 0; }
  }
  long wrow[B_LD];
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const int n = n0 + r0 + 64 * i;
    wrow[i] = n < g.NC ? (long)n * g.Kfull : -1;
  }

  int c = k8, ta = 0, tb = 0, kk = k8;
  long rowoff[A_LD];
  long tapoff = 0;
  constexpr unsigned OOB = 0x80000000u;
  unsigned aoff[A_LD], boff[B_LD], wrow2[B_LD];
  __amdgpu_buffer_rsrc_t rs_a, rs_b, rs_a2, rs_p;
  int n_first = 0, cb = 0;
  if constexpr (BUF) {
    const int ohw = g.OHs * g.OWs;
    long mlast = m0 + BM; if (mlast > g.M) mlast = g.M;
    n_first = (int)(m0 / ohw);
    const int n_last = (int)((mlast - 1) / ohw);
    const unsigned a_bytes = (unsigned)(((long)(n_last - n_first + 1) * g.SH * g.SW - 1) * g.ld_src * 2 + (long)g.SC * 2);
    rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (long)n_first * g.SH * g.SW * g.ld_src), 0, (int)a_bytes, 0x00020000);
    rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)wm, 0, (int)((long)g.NC * g.Kfull * 2), 0x00020000);
    if constexpr (ATR == 2)
      rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)((const __bf16*)F.src2 + (long)n_first * g.SH * g.SW * g.ld_src), 0, (int)a_bytes, 0x00020000);
    if constexpr (ATR != 0)
      rs_p = __builtin_amdgcn_make_buffer_rsrc((void*)F.acoef, 0, 5 * g.SC * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int n = n0 + r0 + 64 * i;
      wrow2[i] = n < g.NC ? (unsigned)n * (unsigned)g.Kfull * 2u : OOB;
    }
  }
  auto retap = [&]() {
    const int kh = g.kh0 + ta * g.kstep, kw = g.kw0 + tb * g.kstep;
    tapoff = (long)(kh * g.KW + kw) * g.SC;
    if constexpr (BUF) {
      const bool kvalid = ta < g.KHs && g.KWs > 0;
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        int sh, sw;
        bool ok = kvalid && rn[i] >= 0;
        if (DGRAD) {
          const int th = rh[i] - kh, tw = rw[i] - kw;
          ok = ok && th >= 0 && tw >= 0;
          sh = th >> g.sshift; sw = tw >> g.sshift;
        } else { sh = rh[i] + kh; sw = rw[i] + kw; }
        ok = ok && (unsigned)sh < (unsigned)g.SH && (unsigned)sw < (unsigned)g.SW;
        const unsigned pix = (unsigned)(((rn[i] - n_first) * g.SH + sh) * g.SW + sw);
        aoff[i] = ok ? pix * (unsigned)(g.ld_src * 2) + (unsigned)(cb + k8) * 2u : OOB;
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i)
        boff[i] = kvalid ? wrow2[i] + (unsigned)((int)tapoff + cb + k8) * 2u : OOB;
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
  retap();
  auto advance = [&]() {
    if constexpr (BUF) {
      cb += HBK;
      if (cb >= g.SC) { cb -= g.SC; if (++tb == g.KWs) { tb = 0; ++ta; } retap(); }
      else {
#pragma unroll
        for (int i = 0; i < A_LD; ++i) aoff[i] += HBK * 2;
#pragma unroll
        for (int i = 0; i < B_LD; ++i) boff[i] += HBK * 2;
      }
      return;
    }
    c += HBK; kk += HBK;
    if (c >= g.SC) { c -= g.SC; if (++tb == g.KWs) { tb = 0; ++ta; } retap(); }
  };

  bf16x8 a_st[A_LD], b_st[B_LD];
  bool a_ok[A_LD], b_ok[B_LD];
  bf16x8 zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = (__bf16)0.f;
  bf16x8 a_st2[ATR == 2 ? A_LD : 1];
  f32x4 tp[ATR == 2 ? 6 : 4];            // per-channel parameters of the staged K tile (8 channels cb+k8 .. +7, two float4 per row)
  auto load_tile = [&]() {
    if constexpr (BUF) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        a_st[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a, (int)aoff[i], 0, 0));
        if constexpr (ATR == 2) a_st2[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_a2, (int)aoff[i], 0, 0));
        if constexpr (ATR != 0 && MASK) a_ok[i] = (int)aoff[i] >= 0;
      }
      if constexpr (ATR != 0) {
        const int po = (cb + k8) * 4;      // fp32 coefficient rows; channels cb+k8 .. +7 (< SC always)
        constexpr int R0 = ATR == 1 ? 2 : 0, R1 = ATR == 1 ? 4 : 1;      // ATR 1: scale, shift2 | ATR 2: A, nK2, (C2 = row 2)
        tp[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po, R0 * g.SC * 4, 0));
        tp[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po + 16, R0 * g.SC * 4, 0));
        tp[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po, R1 * g.SC * 4, 0));
        tp[3] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po + 16, R1 * g.SC * 4, 0));
        if constexpr (ATR == 2) {
          tp[4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po, 2 * g.SC * 4, 0));
          tp[5] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_p, po + 16, 2 * g.SC * 4, 0));
        }
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i)
        b_st[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_b, (int)boff[i], 0, 0));
      return;
    }
    const bool kvalid = kk < g.Ktot;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const bool ok = kvalid && rowoff[i] >= 0;
      a_ok[i] = ok;
      a_st[i] = *reinterpret_cast<const bf16x8*>(src + (ok ? rowoff[i] + c : 0));
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const bool ok = kvalid && wrow[i] >= 0;
      b_ok[i] = ok;
      b_st[i] = *reinterpret_cast<const bf16x8*>(wm + (ok ? wrow[i] + tapoff + c : 0));
    }
  };
  auto transform_tile = [&]() {
    if constexpr (ATR != 0) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float x = (float)a_st[i][e];
          float v;
          if constexpr (ATR == 1) v = fmaxf(__builtin_fmaf(x, tp[e >> 2][e & 3], tp[2 + (e >> 2)][e & 3]), 0.f);
          else v = __builtin_fmaf(tp[2 + (e >> 2)][e & 3], (float)a_st2[i][e], __builtin_fmaf(tp[e >> 2][e & 3], x, tp[4 + (e >> 2)][e & 3]));
          o[e] = (__bf16)v;
        }
        if constexpr (MASK) a_st[i] = a_ok[i] ? o : zero8; else a_st[i] = o;
      }
    }
  };
  auto store_tile = [&](int buf) {
    __bf16* a = As + buf * BM * HLD;
    __bf16* b = Bs + buf * BN * HLD;
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      *reinterpret_cast<bf16x8*>(a + (r0 + 64 * i) * HLD + k8) = (BUF || a_ok[i]) ? a_st[i] : zero8;
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      *reinterpret_cast<bf16x8*>(b + (r0 + 64 * i) * HLD + k8) = (BUF || b_ok[i]) ? b_st[i] : zero8;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int KT = (g.Ktot + HBK - 1) / HBK;
  load_tile();
  transform_tile();
  store_tile(0);
  __syncthreads();

  const int li = lane & 31, lh = lane >> 5;
  for (int kt = 0; kt < KT; ++kt) {
    const int buf = kt & 1;
    const bool more = kt + 1 < KT || (g.flags & GF_KTAIL);        // (wave-uniform) the last tile issues no loads: at K = 64 / 128 -- the short-K layers of
    if (more) {                           // stages 1-2 -- a past-the-end load round and its transform / LDS store were a third /
      advance();                          // a fifth of the loop
      load_tile();                        // next tile's loads first; consumed after the MFMA chain
    }
    __builtin_amdgcn_sched_barrier(0);
    const __bf16* a = As + buf * BM * HLD + (wm0 + li) * HLD + 8 * lh;
    const __bf16* b = Bs + buf * BN * HLD + (wn0 + li) * HLD + 8 * lh;
#pragma unroll
    for (int s = 0; s < HBK / 16; ++s) {
      bf16x8 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * HLD + s * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(b + j * 32 * HLD + s * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (more) {
      transform_tile();
      store_tile(buf ^ 1);
    }
    __syncthreads();
  }

  // ---- vector epilogue through LDS (fp32 staging): a lane owns VW consecutive channels of one pixel.  VW = 8 (16-byte bf16 stores and
  // operand loads) whenever the channel count and the pointers allow: with 8-byte stores the epilogue is store-ISSUE-bound (twice
  // the store instructions for the same bytes; MI355X_MICROARCH.md "epilogue store tail"), and the short-K layers of stages 1-2 are
  // all epilogue.  VW = 4 keeps channel counts that are multiples of 4 only.
  const bool accum = g.flags & GF_ACCUM;
  const bool stats = EPI == 0 && (g.flags & GF_STATS) != 0;
  auto epilogue = [&](auto VW_) {
    constexpr int VW = decltype(VW_)::value;
    constexpr int Q = VW / 4;              // f32x4 quads per lane
    constexpr int SLD = WN + 4;
    constexpr int CV = WN / VW;            // lanes per staged row
    constexpr int RPP2 = 64 / CV;          // rows per pass of the wave
    constexpr int NT = 32 / RPP2;          // row passes per 32-row tile
    typedef __bf16 bf16xv __attribute__((ext_vector_type(VW)));
    float* stage = smem + wave * 32 * SLD;
    const int srow = lane / CV, scv = lane % CV;
    const int n = n0 + wn0 + scv * VW;
    f32x4 kshift[Q], st0[Q], st1[Q], e_scale[Q], e_shift2[Q], e_mean[Q];      // e_mean: plane 1 is sum g*(x - mean), see conv_gemm.hip
#pragma unroll
    for (int q = 0; q < Q; ++q) kshift[q] = st0[q] = st1[q] = e_scale[q] = e_shift2[q] = e_mean[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == 1) {
      if (n < g.NC) {
#pragma unroll
        for (int q = 0; q < Q; ++q) e_mean[q] = *reinterpret_cast<const f32x4*>(F.ep_fcoef + n + 4 * q);
      }
      if (n < g.NC && !F.ep_mask) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          e_scale[q] = *reinterpret_cast<const f32x4*>(F.ep_fcoef + 2 * (long)g.NC + n + 4 * q);
          e_shift2[q] = *reinterpret_cast<const f32x4*>(F.ep_fcoef + 4 * (long)g.NC + n + 4 * q);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // Every global operand of this 32-row tile's epilogue (the accumulate target, the raw tensor of the BatchNorm below and its
      // sign bytes) is requested up front -- NT loads in flight per lane behind the LDS transpose instead of one exposed HBM round
      // trip per row pass (the expanding 1x1 data gradients of stages 1-2, K = 64 / 128, are all epilogue: DESIGN.md section 3b).
      int pixs[NT];                        // destination pixel (< 2^31: checked by the C-ABI launchers), -1 = row / column outside
      bf16xv o_pre[NT], x_pre[NT];
      int mb_pre[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = t * RPP2 + srow;
        const long m = m0 + wm0 + i * 32 + row;
        const bool okt = m < g.M && n < g.NC;
        long pix = m;
        if (DGRAD && g.step > 1 && okt) {
          const int ohw = g.OHs * g.OWs;
          const int nn = (int)(m / ohw);
          const int rem = (int)(m - (long)nn * ohw);
          const int ii = rem / g.OWs, jj = rem - ii * g.OWs;
          pix = ((long)nn * g.OH + g.h0 + ii * g.step) * g.OW + g.w0 + jj * g.step;
        }
        pixs[t] = okt ? (int)pix : -1;
        mb_pre[t] = 0;
        if (okt) {
          if (accum) o_pre[t] = *reinterpret_cast<const bf16xv*>(dst + pix * g.ld_dst + n);
          if constexpr (EPI == 1) {
            x_pre[t] = *reinterpret_cast<const bf16xv*>((const __bf16*)F.ep_x + pix * F.ld_ep + n);
            if (F.ep_mask) {
              const unsigned char* mp = F.ep_mask + pix * (long)(g.NC >> 2) + (n >> 2);
              if constexpr (VW == 8) mb_pre[t] = *reinterpret_cast<const unsigned short*>(mp);     // bytes of channels n..n+3, n+4..n+7
              else mb_pre[t] = *mp;
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * SLD + j * 32 + li] = acc[i][j][r];
      // per-wave staging region and per-wave BatchNorm shift (its own first row): wave-local fences, no workgroup barrier
      // (same scheme as conv_gemm.hip's epilogue; the row halves are re-based when combined below)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (stats && i == 0) {
#pragma unroll
        for (int q = 0; q < Q; ++q) kshift[q] = *reinterpret_cast<const f32x4*>(stage + scv * VW + 4 * q);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = t * RPP2 + srow;
        if (pixs[t] >= 0) {
          bf16xv ov;
#pragma unroll
          for (int q = 0; q < Q; ++q) {
            f32x4 v = *reinterpret_cast<const f32x4*>(stage + row * SLD + scv * VW + 4 * q);
            if (stats) { const f32x4 d = v - kshift[q]; st0[q] += d; st1[q] += d * d; }
            if (accum) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += (float)o_pre[t][4 * q + e];
            }
            if constexpr (EPI == 1) {
              f32x4 xr;
#pragma unroll
              for (int e = 0; e < 4; ++e) xr[e] = (float)x_pre[t][4 * q + e];
              if (F.ep_mask) {
                const int mb = mb_pre[t] >> (8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (mb >> e) & 1 ? v[e] : 0.f;
              } else if (g.flags & GF_EPI_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaf(xr[e], e_scale[q][e], e_shift2[q][e]) > 0.f ? v[e] : 0.f;
              }
              st0[q] += v;
              st1[q] = __builtin_elementwise_fma(v, xr - e_mean[q], st1[q]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) ov[4 * q + e] = (__bf16)v[e];
          }
          *reinterpret_cast<bf16xv*>(dst + (long)pixs[t] * g.ld_dst + n) = ov;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if constexpr (EPI == 1) {
#pragma unroll
      for (int o = 32; o >= CV; o >>= 1) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) { st0[q][e] += __shfl_xor(st0[q][e], o, 64); st1[q][e] += __shfl_xor(st1[q][e], o, 64); }
      }
      float* red = smem + 4 * 32 * SLD;          // [wave][2][WN]
      if (srow == 0) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            red[(wave * 2 + 0) * WN + scv * VW + 4 * q + e] = st0[q][e];
            red[(wave * 2 + 1) * WN + scv * VW + 4 * q + e] = st1[q][e];
          }
      }
      __syncthreads();
      if ((wave >> 1) == 0 && srow == 0 && n < g.NC) {
        float* pp = F.ep_part + ((long)F.ep_chunk0 + tile_m) * 2 * g.NC;
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cc = scv * VW + 4 * q + e;
            pp[n + 4 * q + e] = st0[q][e] + red[((wave + 2) * 2 + 0) * WN + cc];
            pp[g.NC + n + 4 * q + e] = st1[q][e] + red[((wave + 2) * 2 + 1) * WN + cc];
          }
      }
      return;
    }
    if (stats) {
#pragma unroll
      for (int o = 32; o >= CV; o >>= 1) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) { st0[q][e] += __shfl_xor(st0[q][e], o, 64); st1[q][e] += __shfl_xor(st1[q][e], o, 64); }
      }
      float* red = smem + 4 * 32 * SLD;          // [wave][3][WN] = (S1, S2, K)
      if (srow == 0) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            red[(wave * 3 + 0) * WN + scv * VW + 4 * q + e] = st0[q][e];
            red[(wave * 3 + 1) * WN + scv * VW + 4 * q + e] = st1[q][e];
            red[(wave * 3 + 2) * WN + scv * VW + 4 * q + e] = kshift[q][e];
          }
      }
      __syncthreads();
      if ((wave >> 1) == 0 && srow == 0 && n < g.NC) {
        float* pp = g.stat_part + (long)tile_m * 3 * g.NC;
        long nl = g.M - (m0 + WM);
        const float nb = nl <= 0 ? 0.f : (nl > WM ? (float)WM : (float)nl);
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cc = scv * VW + 4 * q + e;
            const float s1b = red[((wave + 2) * 3 + 0) * WN + cc], s2b = red[((wave + 2) * 3 + 1) * WN + cc];
            const float d = red[((wave + 2) * 3 + 2) * WN + cc] - kshift[q][e];
            pp[n + 4 * q + e] = st0[q][e] + (s1b + nb * d);
            pp[g.NC + n + 4 * q + e] = st1[q][e] + (s2b + 2.f * d * s1b + nb * d * d);
            pp[2 * g.NC + n + 4 * q + e] = kshift[q][e];
          }
      }
    }
  };
  const bool wide = (g.flags & GF_EPI_VW4) == 0 && (g.NC & 7) == 0 && (g.ld_dst & 7) == 0 && (((uintptr_t)dst) & 15) == 0 &&
                    (EPI == 0 || ((F.ld_ep & 7) == 0 && (((uintptr_t)F.ep_x) & 15) == 0));
  if (wide) epilogue(std::integral_constant<int, 8>{});
  else epilogue(std::integral_constant<int, 4>{});
}

template <int BN, bool DGRAD, bool BUF, int ATR = 0, int EPI = 0, bool MASK = true>
static int launch_gather_bf16_impl(const __bf16* src, const __bf16* wm, __bf16* dst, const GatherGeom& g, hipStream_t st,
                                   const GatherFuse* fuse = nullptr) {
  const int tiles_m = edrl_cdiv(g.M, 128), tiles_n = edrl_cdiv(g.NC, BN);
  const long nblk = (long)tiles_m * tiles_n;
  if (nblk <= 0) return 0;
  if (nblk > 0x7fffffffL) return EDRL_EINVAL;
  size_t lds = (size_t)2 * (128 + BN) * HLD * sizeof(__bf16);
  const size_t epi = (size_t)(4 * 32 * (BN / 2 + 4) + 4 * 3 * (BN / 2)) * sizeof(float);
  if (epi > lds) lds = epi;
  GatherFuse F;
  if (fuse) F = *fuse; else memset(&F, 0, sizeof(F));
  GatherGeom gm = g;
  const EdrlConfig& cfg = edrl_cfg();
  if (cfg.bf16_epi_vw4 == 1) gm.flags |= GF_EPI_VW4;
  if (cfg.bf16_ktail == 1) gm.flags |= GF_KTAIL;          // (A/B) keep the past-the-end load round
  if constexpr ((ATR == 1 || (ATR == 0 && BUF)) && EPI == 0 && !DGRAD) {
    // (A/B switches EDRL_BF16_FWD_OCC4 / EDRL_BF16_PLAIN_OCC4) plain operands: l3 256->1024 0.53 -> 0.48 ms, l4 512->2048 0.39 -> 0.35 ms
    if (cfg.bf16_fwd_occ4 != 0 && (ATR == 1 || cfg.bf16_plain_occ4 != 0)) {
      auto k4 = conv_gather_bf16_kernel<BN, DGRAD, BUF, ATR, EPI, MASK, true>;
      static bool attr4 = false;
      if (!attr4) { (void)hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr4 = true; }
      hipLaunchKernelGGL(k4, dim3((unsigned)nblk), dim3(256), lds, st, src, wm, dst, gm, tiles_n, F);
      EDRL_LAUNCH_CHECK();
      return 0;
    }
  }
  auto kern = conv_gather_bf16_kernel<BN, DGRAD, BUF, ATR, EPI, MASK>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, src, wm, dst, gm, tiles_n, F);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// EDRL_BF16_V3S: which plain-operand layers take the small-tile LDS-DMA core (conv_bf16_v3s.hip) instead of the 256 x 256 core or
// the register-staged 128-row kernel.  Mode 2: every geometry it can run.  Auto (1): the layers it measured faster on
// (scripts/v3s_layer_bench.py, profiles/r05_v3s_layers_bf16_2048img.txt): 128 output channels with K >= EDRL_V3S_KMIN (1024) --
// the 3x3 layers of the second residual stage, which the 256 x 256 core cannot take (NC % 256) and whose 36 K units amortise the
// one-tile prologue: forward -6 %, data gradient -19 %, with the BatchNorm-backward epilogue -11 %.  On the short-K 1x1 layers it
// LOSES to both existing kernels (`l3 1x1 256->1024` 0.65 ms against 0.50 on the persistent 256 x 256 core): with one tile per
// workgroup the cold prologue of every 8-unit tile is exposed, and two workgroups per CU do not cover it.
static bool gather_bf16_v3s_pick(const GatherGeom& g) {
  const int mode = edrl_cfg().bf16_v3s;
  if (mode == 0 || !gather_bf16_v3s_can(g)) return false;
  if (mode == 2) return true;
  return g.NC == 128 && g.Ktot >= edrl_cfg().v3s_kmin && g.step == 1 && (long)edrl_cdiv(g.M, 128) >= 1024;
}

template <int BN, bool DGRAD>
static int launch_gather_bf16(const __bf16* src, const __bf16* wm, __bf16* dst, const GatherGeom& g, hipStream_t st) {
  const long ohw = (long)g.OHs * g.OWs;
  const bool buf = edrl_cfg().gather_buf != 0 && ohw > 0 && (128 / ohw + 2) * g.SH * g.SW * g.ld_src * 2 < (1L << 31) &&
                   (long)g.NC * g.Kfull * 2 < (1L << 31);   // (rows < 2^31: checked by the extern "C" launchers)
  if (buf) return launch_gather_bf16_impl<BN, DGRAD, true>(src, wm, dst, g, st);
  return launch_gather_bf16_impl<BN, DGRAD, false>(src, wm, dst, g, st);
}

static bool gather_fused_ok_bf16(const GatherGeom& g) {
  const long ohw = (long)g.OHs * g.OWs;
  return (g.SC % HBK == 0) && (g.NC % 4 == 0) && ohw > 0 && (128 / ohw + 2) * g.SH * g.SW * g.ld_src * 2 < (1L << 31) &&
         (long)g.NC * g.Kfull * 2 < (1L << 31);
}
template <bool DGRAD, int ATR, int EPI>
static int dispatch_gather_fused_bf16(const __bf16* src, const __bf16* wm, __bf16* dst, const GatherGeom& g, const GatherFuse& F,
                                      hipStream_t st) {
  if (!gather_fused_ok_bf16(g)) return EDRL_EINVAL;
  const bool mask = !(g.KH == 1 && g.KW == 1 && g.pad == 0);
  if (g.NC <= 64) {
    if (mask) return launch_gather_bf16_impl<64, DGRAD, true, ATR, EPI, true>(src, wm, dst, g, st, &F);
    return launch_gather_bf16_impl<64, DGRAD, true, ATR, EPI, false>(src, wm, dst, g, st, &F);
  }
  if (mask) return launch_gather_bf16_impl<128, DGRAD, true, ATR, EPI, true>(src, wm, dst, g, st, &F);
  return launch_gather_bf16_impl<128, DGRAD, true, ATR, EPI, false>(src, wm, dst, g, st, &F);
}

// ------------------------------------------------------------------------------------------ weight gradient (bf16)
// dW[co][kcol] (fp32) = sum_pixels dY[pix][co] * Xcol[pix][kcol], kcol = (tap, ci): the reduction runs over PIXELS, the
// slow axis of both NHWC operands, while the bf16 MFMA wants 8 consecutive k per lane.  Both tiles are therefore
// staged in their natural [pixel][channel] image (16-byte global loads, channels contiguous) and the fragments are
// fetched with the gfx950 transposing LDS read ds_read_b64_tr_b16 (a 4-pixel x 16-channel block, delivered
// column-major: lane i of a 16-lane group gets channel i of the 4 pixels) — two reads per 8-deep fragment, no scatter.
// Row pitch = BM*2 + 64 bytes puts the 4 rows of a block and the two blocks of a 32-lane half on disjoint banks.
// Split-K over pixels with fp32 partial slabs reduced in fixed order (deterministic), as in the fp32 kernel.
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define WBK 32   // pixels per K tile

struct WgradGeomH {
  long P;
  int OH, OW, Co, SH, SW, SC, KH, KW, stride, pad, Ktot;
  int tiles_per_split;
  int tiles_x, tiles_y;   // 1-D XCD-remapped grid: tiles_x * tiles_y * splits
};

__device__ __forceinline__ bf16x8 tr_frag(const __bf16* base, int pitch, int pix0, int col0, int lane) {
  // block rows pix0..pix0+3 (and +4..+7), columns col0..col0+15; lane 4q+p of the 16-lane group addresses row q, cols 4p..4p+3
  const int g16 = lane & 15, q = g16 >> 2, p4 = g16 & 3;
  const __bf16* a0 = base + (pix0 + q) * pitch + col0 + 4 * p4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * pitch));
  union { struct { s16x4 l, h; } s; bf16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
}

// FASTLD: buffer-descriptor operand loads with a branch-free 32-bit pixel decode (same scheme as conv_gemm.hip's wgrad;
// host-checked: WBK/OW + 1 <= OH, per-workgroup footprints < 2 GiB).
// Fused BatchNorm operands (FASTLD only), as conv_gemm.hip's DYT / XT: DYT 2 forms dY = bf16(A*g + nK2*yraw + C2) from the
// masked gradient (`dy`) and the raw conv output (F.dy2); XT 1 forms X = bf16(relu(x*scale + shift2)) from the raw previous conv
// output; rows outside the split / padding taps are forced to 0 after the transform (MASKX: the layer has padding taps).
struct WgradFuseH {
  const __bf16* dy2;
  const float* bcoef;    // [4][Co]: A, nK2, C2, mean
  const float* xcoef;    // [5][SC]: mean, rstd, scale, shift, shift2
};
template <int BM, int BN, bool FASTLD = false, int DYT = 0, int XT = 0, bool MASKX = true>
__global__ __launch_bounds__(256, 3) void conv_wgrad_bf16_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ x,
                                                                 float* __restrict__ part, WgradGeomH g, WgradFuseH F) {
  static_assert((DYT == 0 && XT == 0) || FASTLD, "operand transforms ride on the buffer-descriptor path");
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int PA = BM + 32, PB = BN + 32;          // LDS row pitch in elements (+64 bytes)
  constexpr int AC8 = BM / 8, BC8 = BN / 8;          // 16-byte chunks per pixel row
  constexpr int A_LD = (WBK * AC8) / 256, B_LD = (WBK * BC8) / 256;
  static_assert(A_LD >= 1 && B_LD >= 1, "tile too small for 256 threads");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __bf16* As = reinterpret_cast<__bf16*>(smem_raw);   // [2][WBK][PA]
  __bf16* Bs = As + 2 * WBK * PA;                     // [2][WBK][PB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm0 = (wave >> 1) * WM, wn0 = (wave & 1) * WN;
  // the workgroups of one split stream the same pixel range: consecutive logical ids, one XCD (one L2)
  const int lid = edrl_xcd_remap(blockIdx.x, gridDim.x);
  const int per_split = g.tiles_x * g.tiles_y;
  const int split = lid / per_split;
  const int trem = lid - split * per_split;
  const int tyi = trem / g.tiles_x;
  const int co0 = tyi * BM, n0 = (trem - tyi * g.tiles_x) * BN;
  const long ptiles = (g.P + WBK - 1) / WBK;
  const long t_begin = (long)split * g.tiles_per_split;
  long t_end = t_begin + g.tiles_per_split;
  if (t_end > ptiles) t_end = ptiles;

  // staging coordinates: chunk (8 channels) fixed per thread, pixel rows advance by WBK per tile
  const int ac8 = tid % AC8, bc8 = tid % BC8;
  const int kcol = n0 + bc8 * 8;
  const bool kvalid = kcol < g.Ktot;
  int kkh = 0, kkw = 0, kc = 0;
  if (kvalid) { const int tap = kcol / g.SC; kc = kcol - tap * g.SC; kkh = tap / g.KW; kkw = tap - kkh * g.KW; }
  const int ohw = g.OH * g.OW;
  int bn_[B_LD], boh[B_LD], bow[B_LD];
  long bp[B_LD], ap[A_LD];
#pragma unroll
  for (int i = 0; i < B_LD; ++i) {
    const long p = t_begin * WBK + (tid + 256 * i) / BC8;
    bp[i] = p;
    const long n = p / ohw;
    const int rem = (int)(p - n * ohw);
    bn_[i] = (int)n; boh[i] = rem / g.OW; bow[i] = rem - boh[i] * g.OW;
  }
#pragma unroll
  for (int i = 0; i < A_LD; ++i) ap[i] = t_begin * WBK + (tid + 256 * i) / AC8;

  bf16x8 a_st[A_LD], b_st[B_LD];
  bool a_ok[A_LD], b_ok[B_LD];
  bf16x8 zero8;
#pragma unroll
  for (int e = 0; e < 8; ++e) zero8[e] = (__bf16)0.f;
  // ---- FASTLD state (byte offsets; >= 2 GiB = masked)
  constexpr unsigned OOBW = 0x80000000u;
  unsigned a_off[A_LD], b_roff[B_LD];
  int b_ih[B_LD], b_iw[B_LD];
  int ih_lim = 0, iw_lim = 0, tapconst = 0, dw_step = 0, dh_step = 0;
  unsigned a_step = 0, c_step = 0, c_wrapw = 0, c_wraph = 0;
  __amdgpu_buffer_rsrc_t rs_dy, rs_x, rs_dy2;
  unsigned dy_last = 0, x_last = 0;
  f32x4 q[DYT == 2 ? 6 : 1];       // A, nK2, C2 of this thread's 8 output channels
  f32x4 xq[XT == 1 ? 4 : 1];       // scale, shift2 of this thread's 8 input channels
  bf16x8 a2_st[DYT == 2 ? A_LD : 1];
  if constexpr (FASTLD) {
    const long p_lo = t_begin * WBK;
    long p_hi = t_end * WBK; if (p_hi > g.P) p_hi = g.P;
    long rows = p_hi - p_lo; if (rows < 1) rows = 1;
    const unsigned ld2y = (unsigned)g.Co * 2u, ld2x = (unsigned)g.SC * 2u;
    rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + p_lo * g.Co), 0, (int)(rows * ld2y), 0x00020000);
    dy_last = (unsigned)(rows * ld2y) - 16u;
    if constexpr (DYT == 2) {
      rs_dy2 = __builtin_amdgcn_make_buffer_rsrc((void*)(F.dy2 + p_lo * g.Co), 0, (int)(rows * ld2y), 0x00020000);
      const int co = co0 + ac8 * 8;
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          q[2 * r + h] = co < g.Co ? *reinterpret_cast<const f32x4*>(F.bcoef + (long)r * g.Co + co + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (XT == 1) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        xq[h] = kvalid ? *reinterpret_cast<const f32x4*>(F.xcoef + 2 * (long)g.SC + kc + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
        xq[2 + h] = kvalid ? *reinterpret_cast<const f32x4*>(F.xcoef + 4 * (long)g.SC + kc + 4 * h) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    const long n_lo = p_lo / ohw, n_hi = (p_hi - 1) / ohw;
    rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)(x + n_lo * g.SH * g.SW * g.SC), 0,
                                             (int)((n_hi - n_lo + 1) * g.SH * g.SW * ld2x), 0x00020000);
    x_last = (unsigned)((n_hi - n_lo + 1) * g.SH * g.SW * ld2x) - 16u;
    const int co = co0 + ac8 * 8;
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      a_off[i] = co < g.Co ? (unsigned)((tid + 256 * i) / AC8) * ld2y + (unsigned)co * 2u : OOBW;
    a_step = WBK * ld2y;
    const int a16 = WBK / g.OW, b16 = WBK - a16 * g.OW;
    c_step = (unsigned)(b16 * g.stride + a16 * g.stride * g.SW) * ld2x;
    c_wrapw = (unsigned)(g.stride * g.SW - g.OW * g.stride) * ld2x;
    c_wraph = (unsigned)(g.SH * g.SW - g.OH * g.stride * g.SW) * ld2x;
    dw_step = b16 * g.stride; dh_step = a16 * g.stride;
    ih_lim = g.OH * g.stride + kkh - g.pad;
    iw_lim = g.OW * g.stride + kkw - g.pad;
    tapconst = ((kkh - g.pad) * g.SW + (kkw - g.pad)) * (int)ld2x + kc * 2;
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      b_ih[i] = boh[i] * g.stride + kkh - g.pad;
      b_iw[i] = bow[i] * g.stride + kkw - g.pad;
      b_roff[i] = (unsigned)(((bn_[i] - (int)n_lo) * g.SH + boh[i] * g.stride) * g.SW + bow[i] * g.stride) * ld2x;
    }
  }
  auto load_tile = [&]() {
    if constexpr (FASTLD) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        a_st[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_dy, (int)a_off[i], 0, 0));
        if constexpr (DYT == 2) {
          a2_st[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_dy2, (int)a_off[i], 0, 0));
          a_ok[i] = a_off[i] <= dy_last;
        }
        a_off[i] += a_step;
      }
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        const bool ok = kvalid && (unsigned)b_ih[i] < (unsigned)g.SH && (unsigned)b_iw[i] < (unsigned)g.SW;
        const unsigned off = ok ? b_roff[i] + (unsigned)tapconst : OOBW;
        b_st[i] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)off, 0, 0));
        if constexpr (XT == 1 && MASKX) b_ok[i] = off <= x_last;
        b_iw[i] += dw_step; b_ih[i] += dh_step; b_roff[i] += c_step;
        const bool w = b_iw[i] >= iw_lim;
        b_iw[i] -= w ? g.OW * g.stride : 0; b_ih[i] += w ? g.stride : 0; b_roff[i] += w ? c_wrapw : 0u;
        const bool h = b_ih[i] >= ih_lim;
        b_ih[i] -= h ? g.OH * g.stride : 0; b_roff[i] += h ? c_wraph : 0u;
      }
      return;
    }
    const int co = co0 + ac8 * 8;
#pragma unroll
    for (int i = 0; i < A_LD; ++i) {
      const bool ok = ap[i] < g.P && co < g.Co;
      a_ok[i] = ok;
      a_st[i] = *reinterpret_cast<const bf16x8*>(dy + (ok ? ap[i] * g.Co + co : 0));
      ap[i] += WBK;
    }
#pragma unroll
    for (int i = 0; i < B_LD; ++i) {
      const int sh = boh[i] * g.stride - g.pad + kkh, sw = bow[i] * g.stride - g.pad + kkw;
      const bool ok = kvalid && bp[i] < g.P && sh >= 0 && sh < g.SH && sw >= 0 && sw < g.SW;
      b_ok[i] = ok;
      b_st[i] = *reinterpret_cast<const bf16x8*>(x + (ok ? (((long)bn_[i] * g.SH + sh) * g.SW + sw) * g.SC + kc : 0));
      bp[i] += WBK; bow[i] += WBK;
      while (bow[i] >= g.OW) { bow[i] -= g.OW; if (++boh[i] == g.OH) { boh[i] = 0; ++bn_[i]; } }
    }
  };
  auto transform_tile = [&]() {
    if constexpr (DYT == 2) {
#pragma unroll
      for (int i = 0; i < A_LD; ++i) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = (__bf16)__builtin_fmaf(q[2 + (e >> 2)][e & 3], (float)a2_st[i][e],
                                        __builtin_fmaf(q[e >> 2][e & 3], (float)a_st[i][e], q[4 + (e >> 2)][e & 3]));
        a_st[i] = a_ok[i] ? o : zero8;
      }
    }
    if constexpr (XT == 1) {
#pragma unroll
      for (int i = 0; i < B_LD; ++i) {
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          o[e] = (__bf16)fmaxf(__builtin_fmaf((float)b_st[i][e], xq[e >> 2][e & 3], xq[2 + (e >> 2)][e & 3]), 0.f);
        if constexpr (MASKX) b_st[i] = b_ok[i] ? o : zero8; else b_st[i] = o;
      }
    }
  };
  auto store_tile = [&](int buf) {
    __bf16* a = As + buf * WBK * PA;
    __bf16* b = Bs + buf * WBK * PB;
#pragma unroll
    for (int i = 0; i < A_LD; ++i)
      *reinterpret_cast<bf16x8*>(a + ((tid + 256 * i) / AC8) * PA + ac8 * 8) = (FASTLD || a_ok[i]) ? a_st[i] : zero8;
#pragma unroll
    for (int i = 0; i < B_LD; ++i)
      *reinterpret_cast<bf16x8*>(b + ((tid + 256 * i) / BC8) * PB + bc8 * 8) = (FASTLD || b_ok[i]) ? b_st[i] : zero8;
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  const int cgrp = 16 * ((lane >> 4) & 1);   // column block of this lane's 16-lane group inside the 32-wide MFMA tile
  if (t_begin < t_end) {
    load_tile();
    transform_tile();
    store_tile(0);
    __syncthreads();
    for (long t = t_begin; t < t_end; ++t) {
      const int buf = (int)((t - t_begin) & 1);
      if (t + 1 < t_end) load_tile();
      __builtin_amdgcn_sched_barrier(0);
      const __bf16* a = As + buf * WBK * PA;
      const __bf16* b = Bs + buf * WBK * PB;
#pragma unroll
      for (int s = 0; s < WBK / 16; ++s) {
        bf16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = tr_frag(a, PA, 16 * s + 8 * lh, wm0 + 32 * i + cgrp, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = tr_frag(b, PB, 16 * s + 8 * lh, wn0 + 32 * j + cgrp, lane);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < t_end) { transform_tile(); store_tile(buf ^ 1); }
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

__global__ __launch_bounds__(256) void splitk_reduce_h_kernel(const float* __restrict__ part, float* __restrict__ dw, long n,
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

static void wgrad_plan_h(long P, int Co, int Ktot, int* splits, int* tiles_per_split) {
  const long tiles = (long)edrl_cdiv(Co, 128) * edrl_cdiv(Ktot, 128);      // (Co <= 64 runs 64-row tiles: the same count)
  const long ptiles = (P + WBK - 1) / WBK;
  // 768 workgroups are resident (256 CUs x 3): aim just under a whole number of rounds (see conv_gemm.hip wgrad_plan)
  const long target = edrl_cfg().wgrad_target_bf16;
  long want = target / tiles;
  if (want < 1) want = 1;
  long max_by_len = ptiles / 16; if (max_by_len < 1) max_by_len = 1;
  long s = want < max_by_len ? want : max_by_len;
  if (s < 1) s = 1;
  if (s > 1024) s = 1024;
  long tps = (ptiles + s - 1) / s;
  s = (ptiles + tps - 1) / tps;
  *splits = (int)(s < 1 ? 1 : s);
  *tiles_per_split = (int)tps;
}

// fp32 -> bf16 (round to nearest even), 4 elements per lane
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out, long n) {
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += (long)gridDim.x * blockDim.x * 4) {
    if (i + 3 < n) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(in + i);
      bf16x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
      *reinterpret_cast<bf16x4*>(out + i) = o;
    } else {
      for (long k = i; k < n; ++k) out[k] = (__bf16)in[k];
    }
  }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const __bf16* __restrict__ in, float* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = (float)in[i];
}
// in fp32 [A][B][C] -> out bf16 [C][B][A]
__global__ void permute_021_bf16_kernel(const float* __restrict__ in, __bf16* __restrict__ out, int A, int B, int C) {
  const long n = (long)A * B * C;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = (int)(i % A);
  const long t = i / A;
  const int b = (int)(t % B);
  const int c = (int)(t / B);
  out[i] = (__bf16)in[((long)a * B + b) * C + c];
}

static inline int cast_grid(long n) {
  long b = (n / 4 + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

extern "C" {

// y (bf16) = conv(x (bf16), w (bf16 [Co,KH,KW,Ci])), optional fused BatchNorm chunk partials (fp32, from the accumulators).
int edrl_conv2d_nhwc_fwd_bf16(const void* x, const void* w, void* y, float* stat_part, size_t stat_part_bytes, int N,
                              int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad,
                              hipStream_t st) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || KH <= 0 || KW <= 0 || stride <= 0 ||
      pad < 0 || (Ci % HBK) || (Co % 4))
    return EDRL_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 7)) return EDRL_EINVAL;
  if ((long)N * Ho * Wo > 0x7fffffffL) return EDRL_EINVAL;
  GatherGeom g;
  g.M = (int)((long)N * Ho * Wo);
  g.OH = Ho; g.OW = Wo; g.NC = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Ci;
  g.ld_src = Ci; g.ld_dst = Co; g.ld_aux = 0; g.flags = 0;
  g.h0 = g.w0 = 0; g.step = 1; g.OHs = Ho; g.OWs = Wo;
  g.kh0 = g.kw0 = 0; g.kstep = 1; g.KHs = KH; g.KWs = KW; g.Kfull = g.Ktot; g.sshift = 0;
  g.stat_part = nullptr; g.stat_shift = nullptr;
  if (stat_part) {
    if (stat_part_bytes < (size_t)(((long)g.M + 127) / 128) * 3 * Co * sizeof(float)) return EDRL_ENOSPC;
    g.flags |= GF_STATS;
    g.stat_part = stat_part;
  }
  if (gather_bf16_v3s_pick(g)) return launch_gather_bf16_v3s(x, w, y, g, false, st);         // HBM-leaning layers: 128x128 LDS-DMA core, 2 per CU
  if (gather_bf16_v3_ok(g, false)) return launch_gather_bf16_v3(x, w, y, g, false, st);      // K-heavy layers: 256x256 LDS-DMA core
  if (Hi == Ho && Wi == Wo && !((uintptr_t)y & 15) && conv1x1_k64_ok(N, Hi, Wi, Ci, Co, KH, KW, stride, pad))
    return launch_conv1x1_k64(x, nullptr, w, y, N, Hi, Wi, Ci, Co, stat_part, st);                    // 64 -> 128..512 1x1: streaming kernel
  if (Hi == Ho && Wi == Wo && !((uintptr_t)y & 15) && conv3x3_c64_ok(N, Hi, Wi, Ci, Co, KH, KW, stride, pad))
    return launch_conv3x3_c64(x, w, 0, y, N, Hi, Wi, stat_part, nullptr, nullptr, nullptr, nullptr, st);   // 64 -> 64 3x3: weight-stationary kernel
  if (Co <= 64) return launch_gather_bf16<64, false>((const __bf16*)x, (const __bf16*)w, (__bf16*)y, g, st);
  return launch_gather_bf16<128, false>((const __bf16*)x, (const __bf16*)w, (__bf16*)y, g, st);
}

// dx (bf16) [+]= conv_transpose(dy (bf16), wt (bf16 [Ci,KH,KW,Co])): parity-class decomposition as in the fp32 kernel
int edrl_conv2d_nhwc_dgrad_bf16(const void* dy, const void* wt, void* dx, int N, int Hi, int Wi, int Ci, int Ho, int Wo,
                                int Co, int KH, int KW, int stride, int pad, int flags, hipStream_t st) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || stride <= 0 || pad < 0 || (Co % HBK) ||
      (Ci % 4))
    return EDRL_EINVAL;
  if (((uintptr_t)dy & 15) || ((uintptr_t)wt & 15) || ((uintptr_t)dx & 7)) return EDRL_EINVAL;
  if ((long)N * Hi * Wi > 0x7fffffffL) return EDRL_EINVAL;
  if (!(flags & GF_ACCUM) && Hi == Ho && Wi == Wo && !((uintptr_t)dx & 15) && conv3x3_c64_ok(N, Hi, Wi, Co, Ci, KH, KW, stride, pad))
    return launch_conv3x3_c64(dy, wt, 1, dx, N, Hi, Wi, nullptr, nullptr, nullptr, nullptr, nullptr, st);
  int sshift = 0;
  while ((1 << sshift) < stride) ++sshift;
  if ((1 << sshift) != stride) return EDRL_EINVAL;
  GatherGeom g;
  g.OH = Hi; g.OW = Wi; g.NC = Ci; g.SH = Ho; g.SW = Wo; g.SC = Co;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Kfull = KH * KW * Co;
  g.ld_src = Co; g.ld_dst = Ci; g.ld_aux = 0; g.flags = flags & GF_ACCUM;
  g.step = stride; g.kstep = stride; g.sshift = sshift;
  g.stat_part = nullptr; g.stat_shift = nullptr;
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
      if (g.Ktot == 0 && (flags & GF_ACCUM)) continue;
      g.M = (int)((long)N * g.OHs * g.OWs);
      const int rc = gather_bf16_v3s_pick(g) ? launch_gather_bf16_v3s(dy, wt, dx, g, true, st)
                     : gather_bf16_v3_ok(g, true) ? launch_gather_bf16_v3(dy, wt, dx, g, true, st)
                     : Ci <= 64 ? launch_gather_bf16<64, true>((const __bf16*)dy, (const __bf16*)wt, (__bf16*)dx, g, st)
                                : launch_gather_bf16<128, true>((const __bf16*)dy, (const __bf16*)wt, (__bf16*)dx, g, st);
      if (rc) return rc;
    }
  return 0;
}

size_t edrl_conv2d_nhwc_wgrad_bf16_workspace_bytes(int N, int Ho, int Wo, int Co, int Ci, int KH, int KW) {
  int splits, tps;
  wgrad_plan_h((long)N * Ho * Wo, Co, KH * KW * Ci, &splits, &tps);
  const size_t a = (size_t)splits * Co * KH * KW * Ci * sizeof(float);
  const size_t b = wgrad_bf16_v3_workspace_bytes(N, Ho, Wo, Co, Ci, KH, KW);     // either core may serve the call
  return a > b ? a : b;
}
}  // extern "C"

template <int DYT, int XT, bool MASKX>
static void launch_wgrad_bf16(const __bf16* dy, const __bf16* x, float* ws, const WgradGeomH& g, long nblk, size_t lds,
                              const WgradFuseH& F, hipStream_t st) {
  auto kern = conv_wgrad_bf16_kernel<128, 128, true, DYT, XT, MASKX>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, st, dy, x, ws, g, F);
}

static bool wgrad_fast_ok_h(int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int tiles_per_split) {
  const long span = (long)tiles_per_split * WBK;
  return (WBK / Wo + 1 <= Ho) && span * Co * 2 < (1L << 31) && (span / ((long)Ho * Wo) + 2) * Hi * Wi * Ci * 2 < (1L << 31);
}

static int wgrad_bf16_impl(const void* dy, const void* x, float* dw, float* workspace, size_t workspace_bytes, int N, int Hi,
                           int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, int accumulate,
                           const WgradFuseH* fuse, hipStream_t st) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || stride <= 0 || pad < 0 || (Co % 8) ||
      (Ci % 8))
    return EDRL_EINVAL;
  if (((uintptr_t)dy & 15) || ((uintptr_t)x & 15)) return EDRL_EINVAL;
  if (!fuse && wgrad_bf16_v3_ok(N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad)) {     // wide plain layers: 256x256 LDS-DMA core
    int splits3 = 0;
    const int rc = launch_wgrad_bf16_v3(dy, x, workspace, workspace_bytes, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, &splits3, st);
    if (rc) return rc;
    const long n3 = (long)Co * KH * KW * Ci;
    hipLaunchKernelGGL(splitk_reduce_h_kernel, dim3(edrl_cdiv(n3, 1024)), dim3(256), 0, st, workspace, dw, n3, splits3, accumulate);
    EDRL_LAUNCH_CHECK();
    return 0;
  }
  WgradGeomH g;
  g.P = (long)N * Ho * Wo;
  g.OH = Ho; g.OW = Wo; g.Co = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Ci;
  int splits;
  wgrad_plan_h(g.P, Co, g.Ktot, &splits, &g.tiles_per_split);
  const size_t need = (size_t)splits * Co * g.Ktot * sizeof(float);
  if (workspace_bytes < need || workspace == nullptr) return EDRL_ENOSPC;
  constexpr int BM = 128, BN = 128;
  const size_t lds = (size_t)2 * WBK * ((BM + 32) + (BN + 32)) * sizeof(__bf16);
  // Co <= 64 (the 3x3 layers of the first residual stage): 64-row tiles -- half of a 128-row tile's MFMAs would multiply zeros
  const bool bm64_env = edrl_cfg().bf16_wgrad_bm64 != 0;
  if (!fuse && Co <= 64 && bm64_env && wgrad_fast_ok_h(Hi, Wi, Ci, Ho, Wo, Co, g.tiles_per_split)) {
    const size_t lds64 = (size_t)2 * WBK * ((64 + 32) + (BN + 32)) * sizeof(__bf16);
    g.tiles_x = edrl_cdiv(g.Ktot, BN); g.tiles_y = 1;
    const long nb = (long)g.tiles_x * splits;
    WgradFuseH F0;
    memset(&F0, 0, sizeof(F0));
    hipLaunchKernelGGL((conv_wgrad_bf16_kernel<64, BN, true>), dim3((unsigned)nb), dim3(256), lds64, st, (const __bf16*)dy,
                       (const __bf16*)x, workspace, g, F0);
    EDRL_LAUNCH_CHECK();
    const long n64 = (long)Co * g.Ktot;
    hipLaunchKernelGGL(splitk_reduce_h_kernel, dim3(edrl_cdiv(n64, 1024)), dim3(256), 0, st, workspace, dw, n64, splits, accumulate);
    EDRL_LAUNCH_CHECK();
    return 0;
  }
  const bool fast_env = edrl_cfg().wgrad_fast != 0;
  const bool fast_ok = wgrad_fast_ok_h(Hi, Wi, Ci, Ho, Wo, Co, g.tiles_per_split);
  const bool fast = fast_env && fast_ok;
  g.tiles_x = edrl_cdiv(g.Ktot, BN); g.tiles_y = edrl_cdiv(Co, BM);
  const long nblk = (long)g.tiles_x * g.tiles_y * splits;
  if (nblk > 0x7fffffffL) return EDRL_EINVAL;
  if (fuse) {
    if (!fast_ok || !fuse->dy2 || (((uintptr_t)fuse->dy2) & 15)) return EDRL_EINVAL;
    const bool maskx = !(KH == 1 && KW == 1 && pad == 0);
    if (fuse->xcoef && maskx) launch_wgrad_bf16<2, 1, true>((const __bf16*)dy, (const __bf16*)x, workspace, g, nblk, lds, *fuse, st);
    else if (fuse->xcoef) launch_wgrad_bf16<2, 1, false>((const __bf16*)dy, (const __bf16*)x, workspace, g, nblk, lds, *fuse, st);
    else launch_wgrad_bf16<2, 0, true>((const __bf16*)dy, (const __bf16*)x, workspace, g, nblk, lds, *fuse, st);
  } else {
    WgradFuseH F0;
    memset(&F0, 0, sizeof(F0));
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_kernel<BM, BN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      (void)hipFuncSetAttribute((const void*)conv_wgrad_bf16_kernel<BM, BN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      attr_set = true;
    }
    if (fast)
      hipLaunchKernelGGL((conv_wgrad_bf16_kernel<BM, BN, true>), dim3((unsigned)nblk), dim3(256), lds, st, (const __bf16*)dy,
                         (const __bf16*)x, workspace, g, F0);
    else
      hipLaunchKernelGGL((conv_wgrad_bf16_kernel<BM, BN, false>), dim3((unsigned)nblk), dim3(256), lds, st, (const __bf16*)dy,
                         (const __bf16*)x, workspace, g, F0);
  }
  EDRL_LAUNCH_CHECK();
  const long n = (long)Co * g.Ktot;
  hipLaunchKernelGGL(splitk_reduce_h_kernel, dim3(edrl_cdiv(n, 1024)), dim3(256), 0, st, workspace, dw, n, splits, accumulate);
  EDRL_LAUNCH_CHECK();
  return 0;
}

extern "C" {
// dw (fp32 [Co,KH,KW,Ci]) [+]= sum_pix dy (bf16) (x) x (bf16).  Co % 8 == 0 and Ci % 8 == 0.
int edrl_conv2d_nhwc_wgrad_bf16(const void* dy, const void* x, float* dw, float* workspace, size_t workspace_bytes, int N,
                                int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad,
                                int accumulate, hipStream_t st) {
  return wgrad_bf16_impl(dy, x, dw, workspace, workspace_bytes, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, accumulate,
                         nullptr, st);
}

// ---- fused-BatchNorm variants of the bf16 trunk (same contracts as the _f32 entry points of conv_gemm.hip; bf16 tensors,
// fp32 coefficient arrays fcoef [5][C] / bcoef [4][C])
int edrl_conv2d_nhwc_wgrad_bn_bf16(const void* g_in, const void* yraw, const float* bcoef, const void* x, const float* x_fcoef,
                                   float* dw, float* workspace, size_t workspace_bytes, int N, int Hi, int Wi, int Ci, int Ho,
                                   int Wo, int Co, int KH, int KW, int stride, int pad, int accumulate, hipStream_t st) {
  if (!g_in || !yraw || !bcoef || !x) return EDRL_EINVAL;
  WgradFuseH F;
  F.dy2 = (const __bf16*)yraw; F.bcoef = bcoef; F.xcoef = x_fcoef;
  return wgrad_bf16_impl(g_in, x, dw, workspace, workspace_bytes, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, accumulate, &F,
                         st);
}

int edrl_conv2d_nhwc_fwd_bnin_stats_bf16(const void* x, const float* in_fcoef, const void* w, void* y, float* stat_part,
                                         size_t stat_part_bytes, int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH,
                                         int KW, int stride, int pad, hipStream_t st) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || KH <= 0 || KW <= 0 || stride <= 0 ||
      pad < 0 || (Ci % HBK) || (Co % 4) || !stat_part || !in_fcoef)
    return EDRL_EINVAL;
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 7)) return EDRL_EINVAL;
  if ((long)N * Ho * Wo > 0x7fffffffL) return EDRL_EINVAL;
  GatherGeom g;
  g.M = (int)((long)N * Ho * Wo);
  if (stat_part_bytes < (size_t)(((long)g.M + 127) / 128) * 3 * Co * sizeof(float)) return EDRL_ENOSPC;
  g.OH = Ho; g.OW = Wo; g.NC = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Ci;
  g.ld_src = Ci; g.ld_dst = Co; g.ld_aux = 0; g.flags = GF_STATS;
  g.h0 = g.w0 = 0; g.step = 1; g.OHs = Ho; g.OWs = Wo;
  g.kh0 = g.kw0 = 0; g.kstep = 1; g.KHs = KH; g.KWs = KW; g.Kfull = g.Ktot; g.sshift = 0;
  g.stat_part = stat_part; g.stat_shift = nullptr;
  if (Hi == Ho && Wi == Wo && !((uintptr_t)y & 15) && conv1x1_k64_ok(N, Hi, Wi, Ci, Co, KH, KW, stride, pad))
    return launch_conv1x1_k64(x, in_fcoef, w, y, N, Hi, Wi, Ci, Co, stat_part, st);
  GatherFuse F;
  memset(&F, 0, sizeof(F));
  F.acoef = in_fcoef;
  return dispatch_gather_fused_bf16<false, 1, 0>((const __bf16*)x, (const __bf16*)w, (__bf16*)y, g, F, st);
}

int edrl_conv2d_nhwc_dgrad_bn_bf16(const void* g_in, const void* yraw, const float* bcoef, const void* wt, void* dx, int N,
                                   int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad,
                                   int flags, const void* ep_raw, const unsigned char* ep_mask, const float* ep_fcoef,
                                   int ep_relu, float* ep_part, size_t ep_part_bytes, hipStream_t st) {
  // yraw == NULL && bcoef == NULL: `g_in` already IS d_raw (materialised by edrl_bn_draw_bf16) -- plain operand load, epilogue only
  // (needs ep_raw: without it this is edrl_conv2d_nhwc_dgrad_bf16)
  const bool plain_in = !yraw && !bcoef;
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ci <= 0 || Ho <= 0 || Wo <= 0 || Co <= 0 || stride <= 0 || pad < 0 || !g_in ||
      (!plain_in && (!yraw || !bcoef)) || (plain_in && !ep_raw) || (Co % HBK) || (Ci % 4))
    return EDRL_EINVAL;
  if (((uintptr_t)g_in & 15) || ((uintptr_t)yraw & 15) || ((uintptr_t)wt & 15) || ((uintptr_t)dx & 7)) return EDRL_EINVAL;
  if ((long)N * Hi * Wi > 0x7fffffffL) return EDRL_EINVAL;
  if (ep_raw && (!ep_fcoef || !ep_part)) return EDRL_EINVAL;
  long chunks = 0;
  for (int ph = 0; ph < stride; ++ph)
    for (int pw = 0; pw < stride; ++pw) {
      const int h0 = ((ph - pad) % stride + stride) % stride, w0 = ((pw - pad) % stride + stride) % stride;
      const long ohs = h0 < Hi ? (Hi - h0 + stride - 1) / stride : 0, ows = w0 < Wi ? (Wi - w0 + stride - 1) / stride : 0;
      chunks += ((long)N * ohs * ows + 127) / 128;
    }
  if (ep_raw && ep_part_bytes < (size_t)chunks * 2 * Ci * sizeof(float)) return EDRL_ENOSPC;
  if (plain_in && ep_mask && !(flags & GF_ACCUM) && Hi == Ho && Wi == Wo && !((uintptr_t)dx & 15) && !((uintptr_t)ep_raw & 15) &&
      conv3x3_c64_ok(N, Hi, Wi, Co, Ci, KH, KW, stride, pad))
    return launch_conv3x3_c64(g_in, wt, 1, dx, N, Hi, Wi, nullptr, ep_raw, ep_mask, ep_part, ep_fcoef, st);
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
  F.src2 = yraw; F.acoef = bcoef;
  if (ep_raw) { F.ep_x = ep_raw; F.ld_ep = Ci; F.ep_mask = ep_mask; F.ep_fcoef = ep_fcoef; F.ep_part = ep_part; }
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
      if (g.Ktot == 0 && (flags & GF_ACCUM) && !ep_raw) continue;
      F.ep_chunk0 = chunk0;
      // wide layers with a materialised d_raw: the 256x256 LDS-DMA core with the masking + partial-sum epilogue
      const bool v3e = plain_in && gather_bf16_v3_ok(g, true) && gather_bf16_v3_epi_ok(g, F) && g.Ktot >= edrl_cfg().v3_epi_kmin;
      const bool v3s = plain_in && ep_raw && gather_bf16_v3_epi_ok(g, F) && gather_bf16_v3s_pick(g);
      const int rc = v3s ? launch_gather_bf16_v3s(g_in, wt, dx, g, true, st, &F)
                     : v3e ? launch_gather_bf16_v3(g_in, wt, dx, g, true, st, &F)
                     : plain_in ? dispatch_gather_fused_bf16<true, 0, 1>((const __bf16*)g_in, (const __bf16*)wt, (__bf16*)dx, g, F, st)
                     : ep_raw ? dispatch_gather_fused_bf16<true, 2, 1>((const __bf16*)g_in, (const __bf16*)wt, (__bf16*)dx, g, F, st)
                              : dispatch_gather_fused_bf16<true, 2, 0>((const __bf16*)g_in, (const __bf16*)wt, (__bf16*)dx, g, F, st);
      if (rc) return rc;
      chunk0 += (g.M + 127) / 128;
    }
  return 0;
}

// Weight gradient of the 1-channel stem on the bf16 matrix pipe: dy bf16 [N,Hs,Ws,64] (d_raw of the stem's BatchNorm), xs the fp32
// space-to-depth image [N,Hs,Ws,4] -> dw fp32 [64][4][4][4] (the folded layout: edrl_stem_weight_fold_f32 dir 1 gathers the 7x7 taps).
size_t edrl_stem_wgrad_s2d_bf16_workspace_bytes(int N, int Hs, int Ws) { return stem_wgrad_s2d_bf16_workspace_bytes(N, Hs, Ws); }
int edrl_stem_wgrad_s2d_bf16(const void* dy, const float* xs, float* dw, float* workspace, size_t workspace_bytes, int N, int Hs, int Ws,
                             hipStream_t st) {
  if (!dy || !xs || !dw || !workspace || !stem_s2d_bf16_ok(N, Hs, Ws)) return EDRL_EINVAL;
  if (workspace_bytes < stem_wgrad_s2d_bf16_workspace_bytes(N, Hs, Ws)) return EDRL_ENOSPC;
  const int rc = launch_stem_wgrad_s2d_bf16(dy, xs, workspace, N, Hs, Ws, st);
  if (rc) return rc;
  hipLaunchKernelGGL(splitk_reduce_h_kernel, dim3(edrl_cdiv(4096, 1024)), dim3(256), 0, st, workspace, dw, 4096L, stem_wgrad_s2d_bf16_splits(N, Hs, Ws), 0);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// Both gradients of an expanding 1x1 layer inside a fused-BatchNorm block in one pass over (g, yraw) (conv1x1_bwd_bf16.hip).
int edrl_conv1x1_k64_bwd_ok_bf16(int N, int H, int W, int Ci, int Co) {
  return (edrl_cfg().bf16_k64_bwd != 0 && conv1x1_k64_bwd_ok(N, H, W, Ci, Co)) ? 1 : 0;
}
long edrl_conv1x1_k64_bwd_chunks(int N, int H, int W) { return conv1x1_k64_bwd_chunks(N, H, W); }
size_t edrl_conv1x1_k64_bwd_workspace_bytes(int N, int H, int W) { return conv1x1_k64_bwd_workspace_bytes(N, H, W); }
int edrl_conv1x1_k64_bwd_bf16(const void* g_in, const void* yraw, const float* bcoef, const void* x2raw, const float* x2_fcoef,
                              const void* wt, void* g2, float* ep_part, size_t ep_part_bytes, float* dw, float* workspace,
                              size_t workspace_bytes, int N, int H, int W, int Ci, int Co, hipStream_t st) {
  if (!g_in || !yraw || !bcoef || !x2raw || !x2_fcoef || !wt || !g2 || !ep_part || !dw || !workspace) return EDRL_EINVAL;
  if (!conv1x1_k64_bwd_ok(N, H, W, Ci, Co)) return EDRL_EINVAL;
  if (ep_part_bytes < (size_t)conv1x1_k64_bwd_chunks(N, H, W) * 2 * Ci * sizeof(float)) return EDRL_ENOSPC;
  if (workspace_bytes < conv1x1_k64_bwd_workspace_bytes(N, H, W)) return EDRL_ENOSPC;
  const int rc = launch_conv1x1_k64_bwd(g_in, yraw, bcoef, x2raw, x2_fcoef, wt, g2, ep_part, workspace, N, H, W, st);
  if (rc) return rc;
  const long n = (long)Co * Ci;
  hipLaunchKernelGGL(splitk_reduce_h_kernel, dim3(edrl_cdiv(n, 1024)), dim3(256), 0, st, workspace, dw, n, conv1x1_k64_bwd_splits(N, H, W), 0);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// The 7x7 / stride 2 / pad 3 stem of a 1-channel bf16 trunk as a bf16-MFMA streaming kernel (conv_c64_bf16.hip, ATR 2).
int edrl_stem_conv_s2d_bf16(const float* xs, const void* w, void* y, float* stat_part, size_t stat_part_bytes, int N, int Hs, int Ws,
                            hipStream_t st) {
  if (!xs || !w || !y || !stem_s2d_bf16_ok(N, Hs, Ws)) return EDRL_EINVAL;
  const long M = (long)N * Hs * Ws;
  if (stat_part && stat_part_bytes < (size_t)((M + 127) / 128) * 3 * 64 * sizeof(float)) return EDRL_ENOSPC;
  return launch_stem_s2d_bf16(xs, w, y, N, Hs, Ws, stat_part, st);
}

int edrl_conv2d_fused_ok_bf16(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0 || (Ci % HBK) || (Co % HBK) || (stride != 1 && stride != 2)) return 0;
  if ((long)N * Hi * Wi > 0x7fffffffL || (long)N * Ho * Wo > 0x7fffffffL) return 0;
  const long kfull = (long)KH * KW * Ci;
  if (!((128 / ((long)Ho * Wo) + 2) * Hi * Wi * Ci * 2 < (1L << 31) && (long)Co * kfull * 2 < (1L << 31))) return 0;
  const long cls = ((long)(Hi + stride - 1) / stride) * ((Wi + stride - 1) / stride);
  if (!((128 / (cls > 0 ? cls : 1) + 3) * Ho * Wo * Co * 2 < (1L << 31))) return 0;
  int splits, tps;
  wgrad_plan_h((long)N * Ho * Wo, Co, (int)kfull, &splits, &tps);
  return wgrad_fast_ok_h(Hi, Wi, Ci, Ho, Wo, Co, tps) ? 1 : 0;
}

int edrl_cast_f32_to_bf16(const float* in, void* out, long n, hipStream_t st) {
  if (n <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(cast_grid(n)), dim3(256), 0, st, in, (__bf16*)out, n);
  EDRL_LAUNCH_CHECK();
  return 0;
}
int edrl_cast_bf16_to_f32(const void* in, float* out, long n, hipStream_t st) {
  if (n <= 0) return EDRL_EINVAL;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(cast_grid(n * 4)), dim3(256), 0, st, (const __bf16*)in, out, n);
  EDRL_LAUNCH_CHECK();
  return 0;
}
// in fp32 [A][B][C] -> out bf16 [C][B][A]   (forward weight -> data-gradient weight, cast on the way)
int edrl_permute_weight_bf16(const float* in, void* out, int A, int B, int C, hipStream_t st) {
  if (A <= 0 || B <= 0 || C <= 0) return EDRL_EINVAL;
  const long n = (long)A * B * C;
  hipLaunchKernelGGL(permute_021_bf16_kernel, dim3(edrl_cdiv(n, 256)), dim3(256), 0, st, in, (__bf16*)out, A, B, C);
  EDRL_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
