// bf16 implicit-GEMM convolution core, small-tile form of generation 3 ("v3s", round 5): 128 x 128 output tile, 4 waves (2 x 2, wave
// tile 64 pixels x 64 channels of v_mfma_f32_16x16x32_bf16), operands global -> LDS by LDS-DMA into a 4-slot ring of 32-deep K
// units (16 KiB per unit: 64 KiB per workgroup), hand-counted waits, one barrier per unit -- the skeleton of conv_bf16_v3.hip with
// half the rows and half the columns, so that TWO workgroups share a CU.
//
// Why it exists (profiles/r05_c2_traffic_by_layer.txt): the HBM-bound members of the bf16 trunk -- 1x1 layers with K = 128 .. 1024 --
// ran at 2.9 - 4.7 TB/s of their algorithmic bytes on both existing kernels for opposite reasons.  The 256 x 256 core has ONE
// workgroup per CU: between two tiles nothing hides the epilogue's stores, the wait for them and the cold first loads of the next
// tile (a K = 256 tile is 3.4 us of MFMA between 6 us of stores and ~3 us of exposed latency).  The 128-row kernel has three
// workgroups per CU but stages its operands through registers: 8 MFMAs per wave and K tile cannot cover a global load, so its K loop
// runs at about one memory latency per tile.  Here the DMA ring keeps three units (48 KiB) per workgroup in flight across the
// barriers AND a second workgroup's K loop runs under the first one's epilogue.
//
// LDS image, swizzle, zero-filled out-of-range pieces: exactly as conv_bf16_v3.hip (instruction j of wave w fills rows
// 64 j + 16 w .. +15 of a unit; slot p of row r holds k chunk p ^ G[(r >> 2) & 3]).
// Pipeline per unit u (ring slot u & 3; all fragments of a unit are read in the second half of the unit before):
//     LDS-DMA A(u+3) piece 0 | 8 MFMAs | LDS-DMA A(u+3) piece 1
//     s_waitcnt vmcnt(6) lgkmcnt(0) ; s_barrier          -> unit u+1 has landed for everyone, slot (u+3)&3's old readers are done
//     ds_read B(u+1), A(u+1) | LDS-DMA B(u+3) piece 0 | 8 MFMAs | LDS-DMA B(u+3) piece 1
// Epilogues (all through a [128 pixels][128 channels] bf16 LDS image, whole 256-byte rows to global memory):
//   EPI 0: plain store / accumulate (+ BatchNorm chunk partials of the forward: one 128-row chunk per tile, the two wave rows
//          re-based and combined through LDS like the 128-row kernel does);
//   EPI 1: the BatchNorm-backward epilogue of the data gradient (accumulate, mask with sign bytes or the recomputed decision,
//          partial sums (sum g, sum g*(x - mean)) of the tile's one chunk -> F.ep_part), the arithmetic of conv_bf16_v3.hip EPI 1.
#include "edrl_common.h"
#include "edrl_config.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "conv_bf16_v3.h"
#include "lds_dma.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define S_BM 128
#define S_BN 128
#define S_BK 32
#define S_ABYTES (S_BM * S_BK * 2)      // 8 KiB
#define S_BBYTES (S_BN * S_BK * 2)      // 8 KiB
#define S_UNIT (S_ABYTES + S_BBYTES)    // 16 KiB
#define S_LDS (4 * S_UNIT)              // 64 KiB ring (the epilogue's 32 KiB image + its reduction scratch live in it afterwards)

__device__ __forceinline__ int v3s_swz(int r) {      // G[(r >> 2) & 3], G = {0, 2, 3, 1} (conv_bf16_v3.hip)
  const int q = (r >> 2) & 3;
  return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1);
}

template <bool DGRAD, int EPI>
__global__ __launch_bounds__(256, 2) void conv_gather_bf16_v3s_kernel(const __bf16* __restrict__ src, const __bf16* __restrict__ wm,
                                                                      __bf16* __restrict__ dst, GatherGeom g, int tiles_n,
                                                                      GatherFuse F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TM = 4, TN = 4;                 // wave tile 64 pixels x 64 channels of 16x16 MFMA tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * 64;
  const int lid = edrl_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const long m0 = (long)tile_m * S_BM;
  const int n0 = tile_n * S_BN;

  // ---- DMA addressing: thread -> row (tid >> 2) + 64 j of both operand units, LDS slot tid & 3 = k chunk (tid & 3) ^ G[row]
  const int srow = tid >> 2;                                   // 0..63
  const int kc8 = (((tid & 3) ^ v3s_swz(srow)) * 8);
  const int ohw = g.OHs * g.OWs;
  const int n_first = (int)(((unsigned long long)(unsigned)m0 * g.mg_ohw) >> g.sh_ohw);
  int pb[2], hw[2];
  const bool lin = g.KH == 1 && g.KW == 1 && g.pad == 0 && g.stride == 1 && g.step == 1 && g.SH == g.OHs && g.SW == g.OWs;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long m = m0 + srow + 64 * j;
    if (lin) {
      pb[j] = m < g.M ? (int)(m - (long)n_first * ohw) : -1;
      hw[j] = (16384 << 16) | 16384;
    } else if (m < g.M) {
      const int n = (int)(((unsigned long long)(unsigned)m * g.mg_ohw) >> g.sh_ohw);
      const int rem = (int)m - n * ohw;
      const int ii = (int)(((unsigned long long)(unsigned)rem * g.mg_ow) >> g.sh_ow), jj = rem - ii * g.OWs;
      const int oh = g.h0 + ii * g.step, ow = g.w0 + jj * g.step;
      int rh, rw;
      if (DGRAD) { rh = oh + g.pad; rw = ow + g.pad; }
      else       { rh = oh * g.stride - g.pad; rw = ow * g.stride - g.pad; }
      pb[j] = (n - n_first) * g.SH * g.SW;
      hw[j] = ((rh + 16384) << 16) | (rw + 16384);
    } else { pb[j] = -1; hw[j] = 0; }
  }
  constexpr unsigned OOB = 0x80000000u;
  long mlast = m0 + S_BM; if (mlast > g.M) mlast = g.M;
  const int n_last = (int)(((unsigned long long)(unsigned)(mlast - 1) * g.mg_ohw) >> g.sh_ohw);
  const unsigned a_bytes = (unsigned)(((long)(n_last - n_first + 1) * g.SH * g.SW - 1) * g.ld_src * 2 + (long)g.SC * 2);
  const v3_i32x4 rs_a = v3_make_srd(src + (long)n_first * g.SH * g.SW * g.ld_src, a_bytes);
  const v3_i32x4 rs_b = v3_make_srd(wm, (unsigned)((long)g.NC * g.Kfull * 2));
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)(lds_ptr_t)smem) + (unsigned)wave * 1024u;
  // weight rows n0 + srow + 64 j: NC % 128 == 0 (host-checked), so every row exists; the 64 j part rides in the scalar offset
  const unsigned wrow0 = (unsigned)(n0 + srow) * (unsigned)g.Kfull * 2u;
  const int wj = 64 * g.Kfull * 2;
  unsigned aoff[2], boff;
  int ta = 0, tb = 0, cb = 0;
  auto retap = [&]() {
    const int kh = g.kh0 + ta * g.kstep, kw = g.kw0 + tb * g.kstep;
    const int tapoff = (kh * g.KW + kw) * g.SC;
    const bool kvalid = ta < g.KHs && g.KWs > 0;      // false past the last tap: the tail pieces of the pipeline read as zeros
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rh = (int)((unsigned)hw[j] >> 16) - 16384, rw = (hw[j] & 0xffff) - 16384;
      int sh, sw;
      bool ok = kvalid && pb[j] >= 0;
      if (DGRAD) {
        const int th = rh - kh, tw = rw - kw;
        ok = ok && th >= 0 && tw >= 0;
        sh = th >> g.sshift; sw = tw >> g.sshift;
      } else { sh = rh + kh; sw = rw + kw; }
      ok = ok && (unsigned)sh < (unsigned)g.SH && (unsigned)sw < (unsigned)g.SW;
      const unsigned pix = (unsigned)(pb[j] + sh * g.SW + sw);
      aoff[j] = ok ? pix * (unsigned)(g.ld_src * 2) + (unsigned)(cb + kc8) * 2u : OOB;
    }
    boff = kvalid ? wrow0 + (unsigned)(tapoff + cb + kc8) * 2u : OOB;
  };
  auto advance = [&]() {
    cb += S_BK;
    if (cb >= g.SC) { cb = 0; if (++tb == g.KWs) { tb = 0; ++ta; } retap(); }
    else {
      aoff[0] += S_BK * 2; aoff[1] += S_BK * 2;       // (an OOB offset stays out of range: 2^31 + a few KiB)
      boff += S_BK * 2;
    }
  };
  auto issueA1 = [&](int slot, int j) {
    v3_dma16(lds0 + (unsigned)slot * S_UNIT + (unsigned)j * 4096u, aoff[j], rs_a, 0);
  };
  auto issueB1 = [&](int slot, int j) {
    v3_dma16(lds0 + (unsigned)slot * S_UNIT + S_ABYTES + (unsigned)j * 4096u, boff, rs_b, j == 0 ? 0 : wj);
  };

  // ---- fragment addressing (bytes inside a unit): row fr (+16 i), k chunk fq at slot fq ^ G[fr]
  const int fr = lane & 15, fq = lane >> 4;
  const int a_rd = (wm0 + fr) * 64 + ((fq ^ v3s_swz(fr)) << 4);
  const int b_rd = S_ABYTES + (wn0 + fr) * 64 + ((fq ^ v3s_swz(fr)) << 4);

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a0[TM], a1[TM], b0[TN], b1[TN];        // fragments of the current / the next unit (the two sets alternate)

  const int KU = g.Ktot / S_BK;
  auto rdA = [&](int slot, bf16x8 (&af)[TM]) {
    const unsigned char* s = smem + slot * S_UNIT + a_rd;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(s + i * 1024);
  };
  auto rdB = [&](int slot, bf16x8 (&bf)[TN]) {
    const unsigned char* s = smem + slot * S_UNIT + b_rd;
#pragma unroll
    for (int i = 0; i < TN; ++i) bf[i] = *reinterpret_cast<const bf16x8*>(s + i * 1024);
  };
  auto mma8 = [&](auto Q_, bf16x8 (&af)[TM], bf16x8 (&bf)[TN]) {      // pixel tiles 2q, 2q+1 against the 4 weight tiles
    constexpr int Q = decltype(Q_)::value;
#pragma unroll
    for (int j = 2 * Q; j < 2 * Q + 2; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[i], af[j], acc[i][j], 0, 0, 0);
  };
  using Q0 = std::integral_constant<int, 0>;
  using Q1 = std::integral_constant<int, 1>;
  auto unit = [&](int u, bf16x8 (&acur)[TM], bf16x8 (&bcur)[TN], bf16x8 (&anxt)[TM], bf16x8 (&bnxt)[TN]) {
    const int nslot = (u + 3) & 3;
    __builtin_amdgcn_sched_barrier(0);
    issueA1(nslot, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma8(Q0{}, acur, bcur);
    __builtin_amdgcn_sched_barrier(0);
    issueA1(nslot, 1);
    __builtin_amdgcn_sched_barrier(0);
    // this wave's pieces of unit u+1 have landed: of the later ones, unit u+2 (4) and the pixel pieces of unit u+3 (2) may fly
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    rdB((u + 1) & 3, bnxt);
    rdA((u + 1) & 3, anxt);
    issueB1(nslot, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma8(Q1{}, acur, bcur);
    __builtin_amdgcn_sched_barrier(0);
    issueB1(nslot, 1);
    advance();
  };

  if (KU > 0) {
    retap();
#pragma unroll
    for (int u = 0; u < 3; ++u) { issueA1(u, 0); issueA1(u, 1); issueB1(u, 0); issueB1(u, 1); advance(); }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // unit 0 landed (this wave's 4 pieces), units 1 and 2 still in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    rdB(0, b0);
    rdA(0, a0);
    int u = 0;
    for (; u + 1 < KU; u += 2) {
      unit(u, a0, b0, a1, b1);
      unit(u + 1, a1, b1, a0, b0);
    }
    if (u < KU) unit(u, a0, b0, a1, b1);
    // the pipeline's tail pieces (zeros into consumed slots) must have landed before the epilogue reuses the LDS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();

  float* scratch = reinterpret_cast<float*>(smem + S_BM * S_BN * 2);      // 32 KiB behind the tile image

  // ---- BatchNorm chunk partials from the fp32 accumulators (forward, GF_STATS).  The tile is ONE 128-row chunk; a wave holds 64 of
  // its rows x 64 channels: shifted sums about the wave's own first row, the lower wave row re-bases the upper one's onto the
  // chunk's first row (the 128-row kernel's formulas, conv_bf16.hip) and writes [chunk][3][NC] = (S1, S2, K).
  const bool stats = !DGRAD && EPI == 0 && (g.flags & GF_STATS);
  f32x4 st_k[TN], st_s1[TN], st_s2[TN];
  if (stats) {
    const long crow0 = m0 + wm0;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      f32x4 kk, s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) kk[e] = __shfl(acc[i][0][e], lane & 48, 64);     // the wave's first row (pixel 0 of tile 0)
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        f32x4 d = acc[i][j] - kk;
        if (crow0 + j * 16 + fr >= g.M) d = f32x4{0.f, 0.f, 0.f, 0.f};
        s1 += d;
        s2 = __builtin_elementwise_fma(d, d, s2);
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
      }
      st_k[i] = kk; st_s1[i] = s1; st_s2[i] = s2;
      if (wm0 != 0 && fr == 0) {                 // upper wave row: hand (S1, S2, K) of channels wn0 + 16 i + 4 fq .. +3 to the lower one
        float* o = scratch + ((wave & 1) * 3) * 64 + i * 16 + 4 * fq;
        *reinterpret_cast<f32x4*>(o) = s1;
        *reinterpret_cast<f32x4*>(o + 64) = s2;
        *reinterpret_cast<f32x4*>(o + 128) = kk;
      }
    }
  }

  // ---- bf16 result -> LDS image [128 pixels][128 channels] (256-byte rows, 16-byte chunk c of row r at slot c ^ (r & 15))
  {
    const int half8 = (fq & 1) * 8;
    unsigned char* wr = smem + (wm0 + fr) * 256 + half8;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int slot = (((wn0 >> 3) + 2 * i + (fq >> 1)) ^ fr) << 4;
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[i][j][e];
        *reinterpret_cast<bf16x4*>(wr + j * 16 * 256 + slot) = o;
      }
    }
  }
  __syncthreads();
  if (stats && wm0 == 0 && fr == 0 && m0 < g.M) {
    float* pp = g.stat_part + (long)tile_m * 3 * (long)g.NC;
    long nl = g.M - (m0 + 64);
    const float nb = nl <= 0 ? 0.f : (nl > 64 ? 64.f : (float)nl);      // valid rows of the upper wave row
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const float* o = scratch + ((wave & 1) * 3) * 64 + i * 16 + 4 * fq;
      const f32x4 s1b = *reinterpret_cast<const f32x4*>(o), s2b = *reinterpret_cast<const f32x4*>(o + 64);
      const f32x4 d = *reinterpret_cast<const f32x4*>(o + 128) - st_k[i];
      const int n = n0 + wn0 + i * 16 + 4 * fq;
      f32x4 r1, r2;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        r1[e] = st_s1[i][e] + (s1b[e] + nb * d[e]);
        r2[e] = st_s2[i][e] + (s2b[e] + 2.f * d[e] * s1b[e] + nb * d[e] * d[e]);
      }
      *reinterpret_cast<f32x4*>(pp + n) = r1;
      *reinterpret_cast<f32x4*>(pp + g.NC + n) = r2;
      *reinterpret_cast<f32x4*>(pp + 2 * (long)g.NC + n) = st_k[i];
    }
  }

  // ---- global phase: thread (rr = tid >> 4, c = tid & 15) moves the 16-byte chunk c of rows 16 it + rr, it = 0..7
  const bool accum = g.flags & GF_ACCUM;
  const int rr = tid >> 4, c = tid & 15;
  const int n = n0 + c * 8;                         // (NC % 128 == 0: every column of the tile exists)
  long pixs[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const long m = m0 + it * 16 + rr;
    long pix = m;
    if (DGRAD && g.step > 1 && m < g.M) {           // parity class of a strided data gradient: scattered destination pixel
      const int nn = (int)(((unsigned long long)(unsigned)m * g.mg_ohw) >> g.sh_ohw);
      const int rem = (int)m - nn * ohw;
      const int ii = (int)(((unsigned long long)(unsigned)rem * g.mg_ow) >> g.sh_ow), jj = rem - ii * g.OWs;
      pix = ((long)nn * g.OH + g.h0 + ii * g.step) * g.OW + g.w0 + jj * g.step;
    }
    pixs[it] = m < g.M ? pix : -1;
  }
  if constexpr (EPI == 1) {
    const int nq = g.NC >> 2;
    const __bf16* epx = reinterpret_cast<const __bf16*>(F.ep_x);
    const bool use_mask = F.ep_mask != nullptr;
    const bool use_relu = !use_mask && (g.flags & GF_EPI_RELU);
    float em[8], esc[8], esh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      em[e] = F.ep_fcoef[n + e];
      esc[e] = use_relu ? F.ep_fcoef[2 * (long)g.NC + n + e] : 0.f;
      esh[e] = use_relu ? F.ep_fcoef[4 * (long)g.NC + n + e] : 0.f;
    }
    float s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s0[e] = s1[e] = 0.f;
    bf16x8 opre[8], xpre[8];
    unsigned mb[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {                // all global operands of the tile requested up front
      mb[it] = 0xffffu;
      if (pixs[it] >= 0) {
        xpre[it] = *reinterpret_cast<const bf16x8*>(epx + pixs[it] * F.ld_ep + n);
        if (accum) opre[it] = *reinterpret_cast<const bf16x8*>(dst + pixs[it] * g.ld_dst + n);
        if (use_mask) mb[it] = *reinterpret_cast<const unsigned short*>(F.ep_mask + pixs[it] * nq + (n >> 2));
      }
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      if (pixs[it] >= 0) {
        const int row = it * 16 + rr;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + row * 256 + ((c ^ rr) << 4));
        bf16x8 ov;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float vf = (float)v[e];
          if (accum) vf += (float)opre[it][e];
          const float xe = (float)xpre[it][e];
          const bool keep = use_mask ? ((mb[it] >> ((e & 3) + 8 * (e >> 2))) & 1u) != 0u :      // two sign bytes, 4 channels each
                                      (use_relu ? __builtin_fmaf(xe, esc[e], esh[e]) > 0.f : true);
          vf = keep ? vf : 0.f;
          s0[e] += vf;
          s1[e] = __builtin_fmaf(vf, xe - em[e], s1[e]);
          ov[e] = (__bf16)vf;
        }
        *reinterpret_cast<bf16x8*>(dst + pixs[it] * g.ld_dst + n) = ov;
      }
    }
    // chunk sums: over the 4 row lanes of a wave (lane bits 4, 5), then over the 4 waves in wave order (deterministic)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s0[e] += __shfl_xor(s0[e], 16, 64); s1[e] += __shfl_xor(s1[e], 16, 64);
      s0[e] += __shfl_xor(s0[e], 32, 64); s1[e] += __shfl_xor(s1[e], 32, 64);
    }
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        scratch[(0 * 4 + wave) * 128 + c * 8 + e] = s0[e];
        scratch[(1 * 4 + wave) * 128 + c * 8 + e] = s1[e];
      }
    }
    __syncthreads();
    {
      const int pl = tid >> 7, ch = tid & 127;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) t += scratch[(pl * 4 + w) * 128 + ch];
      if (m0 < g.M) F.ep_part[((long)F.ep_chunk0 + tile_m) * 2 * g.NC + (long)pl * g.NC + n0 + ch] = t;
    }
  } else {
    bf16x8 opre[8];
    if (accum) {
#pragma unroll
      for (int it = 0; it < 8; ++it)
        if (pixs[it] >= 0) opre[it] = *reinterpret_cast<const bf16x8*>(dst + pixs[it] * g.ld_dst + n);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      if (pixs[it] >= 0) {
        const int row = it * 16 + rr;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + row * 256 + ((c ^ rr) << 4));
        if (accum) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] + (float)opre[it][e]);
        }
        *reinterpret_cast<bf16x8*>(dst + pixs[it] * g.ld_dst + n) = v;
      }
    }
  }
}

// EDRL_BF16_V3S: 0 off | 1 auto | 2 wherever the geometry allows
bool gather_bf16_v3s_can(const GatherGeom& g) {
  const long ohw = (long)g.OHs * g.OWs;
  return (g.SC % S_BK == 0) && (g.NC % S_BN == 0) && (g.ld_dst % 8 == 0) && (g.ld_src % 8 == 0) && ohw > 0 && g.M > 0 &&
         g.M < 0x7fffffff && (S_BM / ohw + 2) * g.SH * g.SW * g.ld_src * 2 < (1L << 31) && (long)g.NC * g.Kfull * 2 < (1L << 31) &&
         (g.Ktot % S_BK == 0);
}

int launch_gather_bf16_v3s(const void* src, const void* wm, void* dst, const GatherGeom& g0, bool dgrad, hipStream_t st,
                           const GatherFuse* fuse) {
  GatherGeom g = g0;
  gather_geom_magic(&g);
  const int tiles_m = edrl_cdiv(g.M, S_BM), tiles_n = edrl_cdiv(g.NC, S_BN);
  const long nblk = (long)tiles_m * tiles_n;
  if (nblk <= 0) return 0;
  if (nblk > 0x7fffffffL) return EDRL_EINVAL;
  if (((uintptr_t)src & 15) || ((uintptr_t)wm & 15) || ((uintptr_t)dst & 15)) return EDRL_EINVAL;
  GatherFuse F;
  memset(&F, 0, sizeof(F));
  static bool attr_set[3] = {false, false, false};
  if (fuse && fuse->ep_x) {
    if (!dgrad || !gather_bf16_v3_epi_ok(g, *fuse)) return EDRL_EINVAL;
    auto ke = conv_gather_bf16_v3s_kernel<true, 1>;
    if (!attr_set[2]) { (void)hipFuncSetAttribute((const void*)ke, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS); attr_set[2] = true; }
    hipLaunchKernelGGL(ke, dim3((unsigned)nblk), dim3(256), S_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, *fuse);
  } else if (dgrad) {
    auto kern = conv_gather_bf16_v3s_kernel<true, 0>;
    if (!attr_set[1]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS); attr_set[1] = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), S_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, F);
  } else {
    auto kern = conv_gather_bf16_v3s_kernel<false, 0>;
    if (!attr_set[0]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS); attr_set[0] = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), S_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, F);
  }
  EDRL_LAUNCH_CHECK();
  return 0;
}
