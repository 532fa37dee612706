// Persistent form of the 256x256 LDS-DMA bf16 implicit-GEMM core (conv_bf16_v3.hip): same tile, ring and K loop, but a workgroup
// walks a list of tiles and the work that used to run with an idle matrix pipe between two tiles is taken off the critical path:
//   * the next tile's row decode and its first three ring units (12 LDS-DMA pieces per wave) are issued BEFORE the current tile's
//     epilogue, so the cold first loads (2-3 k cycles) fly behind the stores;
//   * the epilogue leaves from REGISTERS: v_permlane16_swap trades two accumulator tiles between lane rows so that a lane holds 8
//     consecutive channels (16-byte stores, 64 contiguous bytes per pixel and instruction) -- no LDS image, so the ring is free for
//     the next tile as soon as the last fragment has been read; BatchNorm chunk partials come straight from the accumulators as before;
//   * the row decode uses the host-made magic divisors of GatherGeom (no 64-bit divisions).
// In-kernel stamps of round 3 (profiles/r03_v3_dma_ablation.txt): prologue 9.5 k + epilogue 7.1 k cycles of 119 k per `l3 3x3 256`
// tile with nothing overlapping them (one workgroup per CU).  Tiles are dealt so that the tiles of one XCD stay a contiguous range
// (operand panels shared in that XCD's L2), 32 workgroups per XCD striding through it.
// Forward / data gradient (GF_STATS, GF_ACCUM, strided parity classes) and the data gradient with the BatchNorm-backward epilogue
// (EPI 1, see the template).  EDRL_BF16_V3_PERSIST=0: the one-tile-per-workgroup kernel of conv_bf16_v3.hip instead.
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

#define V3_BM 256
#define V3_BN 256
#define V3_BK 32
#define V3_ABYTES (V3_BM * V3_BK * 2)      // 16 KiB
#define V3_BBYTES (V3_BN * V3_BK * 2)      // 16 KiB
#define V3_UNIT (V3_ABYTES + V3_BBYTES)    // 32 KiB
#define V3_LDS (4 * V3_UNIT)               // 128 KiB ring

__device__ __forceinline__ int v3p_swz(int r) {
  const int q = (r >> 2) & 3;
  return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1);
}

// EPI 1 (data gradient): the BatchNorm-backward epilogue of conv_bf16_v3.hip in registers.  A wave owns one 128-row chunk x 64
// channels, so the chunk partial sums (sum g, sum g*(x - mean)) need no cross-wave step: per lane over its 8 pixel tiles, then
// over the 16 pixel lanes of each lane row -> ep_part [ep_chunk0 + row / 128][2][NC].  (Unlike the LDS-staged epilogue of the
// one-tile kernel, which sums the values after their bf16 staging, this one sums the fp32 values before any rounding -- what
// the 128-row kernel does.)
template <bool DGRAD, int EPI = 0>
__global__ __launch_bounds__(512, 2) void conv_gather_bf16_v3p_kernel(const __bf16* __restrict__ src, const __bf16* __restrict__ wm,
                                                                      __bf16* __restrict__ dst, GatherGeom g, int tiles_n, int ntiles,
                                                                      GatherFuse F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TM = 8, TN = 4;                 // wave tile 128 pixels x 64 channels of 16x16 MFMA tiles
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;
  // ---- tile list of this workgroup: XCD x owns the contiguous tile range [tbase, tbase + tcount), its workgroups stride through it
  const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, wstride = gridDim.x >> 3;
  const int tq = ntiles >> 3, tr = ntiles & 7;
  const int tbase = xcd < tr ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int tcount = tq + (xcd < tr ? 1 : 0);
  int tk = wslot;
  if (tk >= tcount) return;

  const int srow = tid >> 2;                                   // 0..127
  const int kc8 = (((tid & 3) ^ v3p_swz(srow)) * 8);
  const int ohw = g.OHs * g.OWs;
  const bool lin = g.KH == 1 && g.KW == 1 && g.pad == 0 && g.stride == 1 && g.step == 1 && g.SH == g.OHs && g.SW == g.OWs;
  constexpr unsigned OOB = 0x80000000u;
  const v3_i32x4 rs_b = v3_make_srd(wm, (unsigned)((long)g.NC * g.Kfull * 2));
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)(lds_ptr_t)smem) + (unsigned)wave * 1024u;
  const int wj = 128 * g.Kfull * 2;

  // ---- state of the tile being LOADED
  long m0 = 0;
  int n0 = 0;
  int pb[2], hw[2];
  v3_i32x4 rs_a;
  unsigned wrow0 = 0;
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
    cb += V3_BK;
    if (cb >= g.SC) { cb = 0; if (++tb == g.KWs) { tb = 0; ++ta; } retap(); }
    else {
      aoff[0] += V3_BK * 2; aoff[1] += V3_BK * 2;       // (an OOB offset stays out of range: 2^31 + a few KiB)
      boff += V3_BK * 2;
    }
  };
  // decode of tile `lid` (logical id = tile_m * tiles_n + tile_n): rows srow + 128 j of both operand units, descriptors, first tap
  auto setup = [&](int lid) {
    const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
    m0 = (long)tile_m * V3_BM;
    n0 = tile_n * V3_BN;
    const int n_first = (int)(((unsigned long long)(unsigned)m0 * g.mg_ohw) >> g.sh_ohw);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long m = m0 + srow + 128 * j;
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
    long mlast = m0 + V3_BM; if (mlast > g.M) mlast = g.M;
    const int n_last = (int)(((unsigned long long)(unsigned)(mlast - 1) * g.mg_ohw) >> g.sh_ohw);
    const unsigned a_bytes = (unsigned)(((long)(n_last - n_first + 1) * g.SH * g.SW - 1) * g.ld_src * 2 + (long)g.SC * 2);
    rs_a = v3_make_srd(src + (long)n_first * g.SH * g.SW * g.ld_src, a_bytes);
    wrow0 = (unsigned)(n0 + srow) * (unsigned)g.Kfull * 2u;
    ta = 0; tb = 0; cb = 0;
    retap();
  };
  auto issueA = [&](int slot) {
    const unsigned base = lds0 + (unsigned)slot * V3_UNIT;
    v3_dma16(base, aoff[0], rs_a, 0);
    v3_dma16(base + 8192, aoff[1], rs_a, 0);
  };
  auto issueB = [&](int slot) {
    const unsigned base = lds0 + (unsigned)slot * V3_UNIT + V3_ABYTES;
    v3_dma16(base, boff, rs_b, 0);
    v3_dma16(base + 8192, boff, rs_b, wj);
  };
  auto issueA1 = [&](int slot, int j) {
    v3_dma16(lds0 + (unsigned)slot * V3_UNIT + (unsigned)j * 8192u, aoff[j], rs_a, 0);
  };
  auto issueB1 = [&](int slot, int j) {
    v3_dma16(lds0 + (unsigned)slot * V3_UNIT + V3_ABYTES + (unsigned)j * 8192u, boff, rs_b, j == 0 ? 0 : wj);
  };

  // ---- fragment addressing (bytes inside a unit): row fr (+16 i), k chunk fq at slot fq ^ G[fr]
  const int fr = lane & 15, fq = lane >> 4;
  const int a_rd = (wm0 + fr) * 64 + ((fq ^ v3p_swz(fr)) << 4);
  const int b_rd = V3_ABYTES + (wn0 + fr) * 64 + ((fq ^ v3p_swz(fr)) << 4);

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ac[4], an[4], bc[TN], bn[TN];

  const int KU = g.Ktot / V3_BK;
  auto rdA = [&](int slot, int mh, bf16x8 (&af)[4]) {
    const unsigned char* s = smem + slot * V3_UNIT + a_rd + mh * 4 * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(s + i * 1024);
  };
  auto rdB = [&](int slot, bf16x8 (&bf)[TN]) {
    const unsigned char* s = smem + slot * V3_UNIT + b_rd;
#pragma unroll
    for (int i = 0; i < TN; ++i) bf[i] = *reinterpret_cast<const bf16x8*>(s + i * 1024);
  };
  auto mma8 = [&](auto MH_, auto Q_, bf16x8 (&af)[4], bf16x8 (&bf)[TN]) {
    constexpr int MH = decltype(MH_)::value, Q = decltype(Q_)::value;
#pragma unroll
    for (int j = 2 * Q; j < 2 * Q + 2; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i)
        acc[i][MH * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[i], af[j], acc[i][MH * 4 + j], 0, 0, 0);
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  // one unit of the K loop, exactly as in conv_bf16_v3.hip (pieces of unit u+3 spread over the unit, one barrier per unit)
  auto unit = [&](int u, bf16x8 (&bcur)[TN], bf16x8 (&bnxt)[TN]) {
    const int slot = u & 3, nslot = (u + 3) & 3;
    __builtin_amdgcn_sched_barrier(0);
    rdA(slot, 1, an);
    issueA1(nslot, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H0{}, H0{}, ac, bcur);
    __builtin_amdgcn_sched_barrier(0);
    issueA1(nslot, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H0{}, H1{}, ac, bcur);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    rdB((u + 1) & 3, bnxt);
    rdA((u + 1) & 3, 0, ac);
    issueB1(nslot, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H1{}, H0{}, an, bcur);
    __builtin_amdgcn_sched_barrier(0);
    issueB1(nslot, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H1{}, H1{}, an, bcur);
    __builtin_amdgcn_sched_barrier(0);
    advance();
  };

  // ---- epilogue of the tile at (em0, en0), from registers
  const bool even = (fq & 1) == 0;
  const int cb0 = even ? 4 * fq : 16 + 4 * (fq - 1);        // after the lane-row swap: this lane's 8 consecutive channels of a 32-channel pair
  auto epilogue = [&](long em0, int en0) {
    if (!DGRAD && (g.flags & GF_STATS)) {                    // BatchNorm chunk partials: this wave owns one 128-row chunk x 64 channels
      const long crow0 = em0 + wm0;
      if (crow0 < g.M) {
        const bool full = crow0 + 128 <= g.M;
        float* pp = g.stat_part + (crow0 >> 7) * 3 * (long)g.NC;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          f32x4 kk, s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int e = 0; e < 4; ++e) kk[e] = __shfl(acc[i][0][e], lane & 48, 64);     // the chunk's first row (pixel 0 of tile 0)
#pragma unroll
          for (int j = 0; j < TM; ++j) {
            f32x4 d = acc[i][j] - kk;
            if (!full) { if (crow0 + j * 16 + fr >= g.M) d = f32x4{0.f, 0.f, 0.f, 0.f}; }
            s1 += d;
            s2 = __builtin_elementwise_fma(d, d, s2);
          }
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
          }
          const int n = en0 + wn0 + i * 16 + 4 * fq;
          if (fr == 0) {
            *reinterpret_cast<f32x4*>(pp + n) = s1;
            *reinterpret_cast<f32x4*>(pp + g.NC + n) = s2;
            *reinterpret_cast<f32x4*>(pp + 2 * (long)g.NC + n) = kk;
          }
        }
      }
    }
    const bool accum = g.flags & GF_ACCUM;
    // EPI 1 operands of this lane's 2 x 8 channels
    const __bf16* epx = reinterpret_cast<const __bf16*>(F.ep_x);
    const bool use_mask = EPI == 1 && F.ep_mask != nullptr;
    const bool use_relu = EPI == 1 && !use_mask && (g.flags & GF_EPI_RELU);
    const int nq = g.NC >> 2;
    f32x4 s0[EPI == 1 ? 4 : 1], s1[EPI == 1 ? 4 : 1];        // [2 p + h]: sums of channels 32 p + cb0 + 4 h .. +3
    if constexpr (EPI == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q) s0[q] = s1[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // destination pixel of each of the lane's 8 rows (a strided class scatters them)
    long pixs[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const long m = em0 + wm0 + 16 * j + fr;
      long pix = m;
      if (DGRAD && g.step > 1 && m < g.M) {                   // parity class of a strided data gradient: scattered destination pixel
        const int nn = (int)(((unsigned long long)(unsigned)m * g.mg_ohw) >> g.sh_ohw);
        const int rem = (int)m - nn * ohw;
        const int ii = (int)(((unsigned long long)(unsigned)rem * g.mg_ow) >> g.sh_ow), jj = rem - ii * g.OWs;
        pix = ((long)nn * g.OH + g.h0 + ii * g.step) * g.OW + g.w0 + jj * g.step;
      }
      pixs[j] = m < g.M ? pix : -1;
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int chan = en0 + wn0 + 32 * p + cb0;
      f32x4 mu0 = {0.f, 0.f, 0.f, 0.f}, mu1 = mu0, sc0 = mu0, sc1 = mu0, sh0 = mu0, sh1 = mu0;
      if constexpr (EPI == 1) {
        mu0 = *reinterpret_cast<const f32x4*>(F.ep_fcoef + chan); mu1 = *reinterpret_cast<const f32x4*>(F.ep_fcoef + chan + 4);
        if (use_relu) {
          sc0 = *reinterpret_cast<const f32x4*>(F.ep_fcoef + 2 * (long)g.NC + chan); sc1 = *reinterpret_cast<const f32x4*>(F.ep_fcoef + 2 * (long)g.NC + chan + 4);
          sh0 = *reinterpret_cast<const f32x4*>(F.ep_fcoef + 4 * (long)g.NC + chan); sh1 = *reinterpret_cast<const f32x4*>(F.ep_fcoef + 4 * (long)g.NC + chan + 4);
        }
      }
#pragma unroll
      for (int jh = 0; jh < 2; ++jh) {
        // the global operands of 4 rows requested together (read-modify-write target, raw tensor, sign bytes)
        bf16x8 oldv[4], xv[4];
        unsigned mb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const long pix = pixs[4 * jh + q];
          mb[q] = 0xffffu;
          if (pix >= 0) {
            if (accum) oldv[q] = *reinterpret_cast<const bf16x8*>(dst + pix * g.ld_dst + chan);
            if constexpr (EPI == 1) {
              xv[q] = *reinterpret_cast<const bf16x8*>(epx + pix * F.ld_ep + chan);
              if (use_mask) mb[q] = *reinterpret_cast<const unsigned short*>(F.ep_mask + pix * nq + (chan >> 2));
            }
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = 4 * jh + q;
          f32x4 v0, v1;
          // v_permlane16_swap: the odd 16-lane rows of the first register trade places with the even rows of the second -> 8
          // consecutive channels per lane (conv_c64_bf16.hip's epilogue idiom; same C/D layout: weight fragment first)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2 * p][j][e]), __float_as_uint(acc[2 * p + 1][j][e]), false, false);
            v0[e] = __uint_as_float(sw[0]); v1[e] = __uint_as_float(sw[1]);
          }
          const long pix = pixs[j];
          if (pix >= 0) {
            if (accum) {
#pragma unroll
              for (int e = 0; e < 4; ++e) { v0[e] += (float)oldv[q][e]; v1[e] += (float)oldv[q][4 + e]; }
            }
            if constexpr (EPI == 1) {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float x0 = (float)xv[q][e], x1 = (float)xv[q][4 + e];
                const bool k0 = use_mask ? ((mb[q] >> e) & 1u) != 0u : (use_relu ? __builtin_fmaf(x0, sc0[e], sh0[e]) > 0.f : true);
                const bool k1 = use_mask ? ((mb[q] >> (8 + e)) & 1u) != 0u : (use_relu ? __builtin_fmaf(x1, sc1[e], sh1[e]) > 0.f : true);
                v0[e] = k0 ? v0[e] : 0.f;
                v1[e] = k1 ? v1[e] : 0.f;
                s0[2 * p][e] += v0[e]; s1[2 * p][e] = __builtin_fmaf(v0[e], x0 - mu0[e], s1[2 * p][e]);
                s0[2 * p + 1][e] += v1[e]; s1[2 * p + 1][e] = __builtin_fmaf(v1[e], x1 - mu1[e], s1[2 * p + 1][e]);
              }
            }
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) { o[e] = (__bf16)v0[e]; o[4 + e] = (__bf16)v1[e]; }
            *reinterpret_cast<bf16x8*>(dst + pix * g.ld_dst + chan) = o;
          }
        }
      }
    }
    if constexpr (EPI == 1) {
      // sum over the 16 pixel lanes of this lane row (xor 1, 2, 4, 8 stay inside it); lane fr == 0 writes its 16 channels
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int o = 1; o < 16; o <<= 1) { s0[q][e] += __shfl_xor(s0[q][e], o, 64); s1[q][e] += __shfl_xor(s1[q][e], o, 64); }
        }
      const long crow0 = em0 + wm0;
      if (fr == 0 && crow0 < g.M) {
        float* pp = F.ep_part + ((long)F.ep_chunk0 + (crow0 >> 7)) * 2 * g.NC;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int chan = en0 + wn0 + 32 * p + cb0 + 4 * h;
            *reinterpret_cast<f32x4*>(pp + chan) = s0[2 * p + h];
            *reinterpret_cast<f32x4*>(pp + g.NC + chan) = s1[2 * p + h];
          }
      }
    }
  };

  // ---- EPI 2: the BatchNorm-backward epilogue with LDS-staged rows.  The tile leaves the accumulators through the LDS image of the
  // one-tile kernel (512-byte pixel rows), but every thread pulls its 16 row pieces into REGISTERS at once, which frees the ring for
  // the next tile's first units before the global phase (raw tensor, accumulate target, sign bytes in, masked gradient out) starts.
  bf16x8 ev[EPI == 2 ? 16 : 1];
  auto stage_epi = [&]() {
    if constexpr (EPI == 2) {
      const int half8 = (fq & 1) * 8;
      unsigned char* wr = smem + (wm0 + fr) * 512 + half8;
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const int slot = (((wn0 >> 3) + 2 * i + (fq >> 1)) ^ fr) << 4;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (__bf16)acc[i][j][e];
          *reinterpret_cast<bf16x4*>(wr + j * 16 * 512 + slot) = o;
        }
      }
      __syncthreads();
      const int rr = tid >> 5, c = tid & 31;
#pragma unroll
      for (int it = 0; it < 16; ++it) ev[it] = *reinterpret_cast<const bf16x8*>(smem + (it * 16 + rr) * 512 + ((c ^ rr) << 4));
      __syncthreads();                              // image consumed: the ring may be refilled
    }
  };
  auto finish_epi = [&](long em0, int en0) {
    if constexpr (EPI == 2) {
      const bool accum = g.flags & GF_ACCUM;
      const int rr = tid >> 5, c = tid & 31;
      const int n = en0 + c * 8;
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
      float s0[2][8], s1[2][8];
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) s0[h][e] = s1[h][e] = 0.f;
#pragma unroll
      for (int h = 0; h < 2; ++h) {                 // chunk h of the tile = rows 128 h .. 128 h + 127
        long pixs[8];
        bf16x8 opre[8], xpre[8];
        unsigned mb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {               // all global operands of the chunk requested up front
          const long m = em0 + (h * 8 + i) * 16 + rr;
          long pix = m;
          if (g.step > 1 && m < g.M) {
            const int nn = (int)(((unsigned long long)(unsigned)m * g.mg_ohw) >> g.sh_ohw);
            const int rem = (int)m - nn * ohw;
            const int ii = (int)(((unsigned long long)(unsigned)rem * g.mg_ow) >> g.sh_ow), jj = rem - ii * g.OWs;
            pix = ((long)nn * g.OH + g.h0 + ii * g.step) * g.OW + g.w0 + jj * g.step;
          }
          pixs[i] = m < g.M ? pix : -1;
          mb[i] = 0xffffu;
          if (pixs[i] >= 0) {
            xpre[i] = *reinterpret_cast<const bf16x8*>(epx + pix * F.ld_ep + n);
            if (accum) opre[i] = *reinterpret_cast<const bf16x8*>(dst + pix * g.ld_dst + n);
            if (use_mask) mb[i] = *reinterpret_cast<const unsigned short*>(F.ep_mask + pix * nq + (n >> 2));
          }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          if (pixs[i] >= 0) {
            const bf16x8 v = ev[EPI == 2 ? h * 8 + i : 0];
            bf16x8 ov;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              float vf = (float)v[e];
              if (accum) vf += (float)opre[i][e];
              const float xe = (float)xpre[i][e];
              const bool keep = use_mask ? ((mb[i] >> ((e & 3) + 8 * (e >> 2))) & 1u) != 0u : (use_relu ? __builtin_fmaf(xe, esc[e], esh[e]) > 0.f : true);
              vf = keep ? vf : 0.f;
              s0[h][e] += vf;
              s1[h][e] = __builtin_fmaf(vf, xe - em[e], s1[h][e]);
              ov[e] = (__bf16)vf;
            }
            *reinterpret_cast<bf16x8*>(dst + pixs[i] * g.ld_dst + n) = ov;
          }
        }
      }
      // chunk sums across the 8 waves through ring slot 3 (free until the next tile's first K unit issues its pieces into it)
      float* red = reinterpret_cast<float*>(smem + 3 * V3_UNIT);    // [chunk][plane][wave][256 channels] = 32 KiB
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s0[h][e] += __shfl_xor(s0[h][e], 32, 64); s1[h][e] += __shfl_xor(s1[h][e], 32, 64); }
      if (lane < 32) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            red[((h * 2 + 0) * 8 + wave) * 256 + c * 8 + e] = s0[h][e];
            red[((h * 2 + 1) * 8 + wave) * 256 + c * 8 + e] = s1[h][e];
          }
      }
      __syncthreads();
      for (int o = tid; o < 1024; o += 512) {       // ordered sum over the 8 waves: deterministic
        const int h = o >> 9, pl = (o >> 8) & 1, ch = o & 255;
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) t += red[((h * 2 + pl) * 8 + w) * 256 + ch];
        if (em0 + h * 128 < g.M) F.ep_part[((long)F.ep_chunk0 + (em0 >> 7) + h) * 2 * g.NC + (long)pl * g.NC + en0 + ch] = t;
      }
    }
  };

  if (wave >= 4) __builtin_amdgcn_s_setprio(1);               // static priority for the second-dispatched half (as conv_bf16_v3.hip)
  setup(tbase + tk);
  if (KU > 0) {
#pragma unroll
    for (int u = 0; u < 3; ++u) { issueA(u); issueB(u); advance(); }
  }
  for (;;) {
    const long em0 = m0;
    const int en0 = n0;
    if (KU > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // units 0..2 of this tile (and the previous tile's stores) have landed
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      rdB(0, bc);
      rdA(0, 0, ac);
      int u = 0;
      for (; u + 1 < KU; u += 2) {
        unit(u, bc, bn);
        unit(u + 1, bn, bc);
      }
      if (u < KU) unit(u, bc, bn);
      // the pipeline's tail pieces (zeros into consumed slots) must have landed before the next tile's first units go into the ring
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    tk += wstride;
    const bool more = tk < tcount;
    __builtin_amdgcn_s_barrier();                              // every wave has read its last fragments: the ring is free
    stage_epi();
    if (more) {
      setup(tbase + tk);
      if (KU > 0) {
#pragma unroll
        for (int u = 0; u < 3; ++u) { issueA(u); issueB(u); advance(); }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (EPI == 2) finish_epi(em0, en0); else epilogue(em0, en0);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (!more) break;
  }
}

// EDRL_BF16_V3_PERSIST: 0 off; 1 forward / plain data gradient only (the data gradient with the BatchNorm-backward epilogue stays on
// the one-tile kernel); 2 also that data gradient, with LDS-staged rows pulled into registers before the next tile's loads (EPI 2);
// 3 the same with the register-form epilogue (EPI 1).  Measured at C2 (6 steps, same box): mode 1 320.4 ms per step, mode 3 324.9 --
// the register epilogue touches 64-byte segments of 16 pixel rows per instruction on THREE streams there (raw tensor, accumulate
// target, result), the LDS-staged epilogue whole 512-byte rows; with one stream (plain stores) the overlap with the next tile's
// loads wins.
bool gather_bf16_v3p_ok(const GatherGeom& g, bool epi) {
  return edrl_cfg().bf16_v3_persist >= (epi ? 2 : 1) && g.M < 0x7fffffff;
}

int launch_gather_bf16_v3p(const void* src, const void* wm, void* dst, const GatherGeom& g0, bool dgrad, hipStream_t st,
                           const GatherFuse* fuse) {
  GatherGeom g = g0;
  gather_geom_magic(&g);
  const int tiles_m = edrl_cdiv(g.M, V3_BM), tiles_n = edrl_cdiv(g.NC, V3_BN);
  const long nt = (long)tiles_m * tiles_n;
  if (nt <= 0) return 0;
  if (nt > 0x7fffffffL) return EDRL_EINVAL;
  if (((uintptr_t)src & 15) || ((uintptr_t)wm & 15) || ((uintptr_t)dst & 15)) return EDRL_EINVAL;
  int grid = nt >= 256 ? 256 : (int)((nt + 7) / 8 * 8);      // one workgroup per CU; a multiple of 8 (XCD-contiguous tile ranges)
  static bool attr_set[2] = {false, false};
  GatherFuse F;
  memset(&F, 0, sizeof(F));
  if (fuse && fuse->ep_x) {
    if (!dgrad) return EDRL_EINVAL;
    static bool attr_e[2] = {false, false};
    if (edrl_cfg().bf16_v3_persist == 3) {          // register form of the epilogue (A/B: loses, see gather_bf16_v3p_ok)
      auto kern = conv_gather_bf16_v3p_kernel<true, 1>;
      if (!attr_e[0]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr_e[0] = true; }
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), V3_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, (int)nt, *fuse);
    } else {                                        // LDS-staged rows, pulled into registers before the next tile's loads are issued
      auto kern = conv_gather_bf16_v3p_kernel<true, 2>;
      if (!attr_e[1]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr_e[1] = true; }
      hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), V3_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, (int)nt, *fuse);
    }
    EDRL_LAUNCH_CHECK();
    return 0;
  }
  if (dgrad) {
    auto kern = conv_gather_bf16_v3p_kernel<true>;
    if (!attr_set[1]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr_set[1] = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), V3_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, (int)nt, F);
  } else {
    auto kern = conv_gather_bf16_v3p_kernel<false>;
    if (!attr_set[0]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr_set[0] = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), V3_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, (int)nt, F);
  }
  EDRL_LAUNCH_CHECK();
  return 0;
}
