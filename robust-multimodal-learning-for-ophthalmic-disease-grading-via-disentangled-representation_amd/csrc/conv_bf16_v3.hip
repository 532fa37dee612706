// bf16 implicit-GEMM convolution core, generation 3: 256 x 256 output tile, 8 waves, v_mfma_f32_16x16x32_bf16, operands moved
// global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds: no VGPR staging, no ds_write), 32-deep K units in a 4-slot LDS ring
// whose loads stay in flight across the workgroup barriers (counted by hand: while unit u is multiplied, units u+1 .. u+3 are
// on their way).  Serves the K-heavy layers of the bf16 ResNet trunks (3x3 convs and wide 1x1 convs with >= 256 output channels;
// forward and data gradient); the 128-row kernel of conv_bf16.hip keeps the
// HBM-bound layers, where three small workgroups per CU hide the epilogue better than one large one.
//
// Contraction: D[n][m] = sum_k W[n][k] * X[m][k], m = output pixel, n = output channel, k = (tap, channel): the WEIGHT fragment is
// the MFMA's first operand, so a lane ends up with 4 consecutive channels of one pixel (C/D map: row = 4*(lane>>4)+reg = channel,
// col = lane&15 = pixel) and the bf16 result is staged to LDS with 8-byte writes, then stored as whole 512-byte pixel rows.
//
// LDS image of one operand unit: [256 rows][32 k] bf16 = 64-byte rows (16 KiB), filled in exactly the order the DMA writes it
// (instruction j of wave w -> rows 128 j + 16 w .. +15, lane l -> row l>>2, 16-byte slot l&3).  Bank conflicts of the fragment
// reads (ds_read_b128: 16 rows at one k chunk per 16-lane group) are removed by an XOR swizzle applied on the SOURCE side: slot p of
// row r holds k chunk p ^ G[(r>>2)&3], G = {0,2,3,1}, and the reader asks for slot chunk ^ G[(row>>2)&3] (conflict-free for the
// hardware's lane groups; checked by enumeration).  Padding taps and rows >= M are out-of-range buffer offsets: the hardware
// range check makes the DMA write zeros (scripts/microbench/ldsdma_oob.hip).
//
// Pipeline per unit u (ring slot u & 3), pixel fragments double-buffered by halves (h0 = pixels 0..63, h1 = 64..127 of the wave):
//     ds_read A(u,h1) | LDS-DMA unit u+3 -> slot (u+3)&3 (4 pieces per wave) | MFMA (u,h0)
//     s_waitcnt vmcnt(8) lgkmcnt(0)  -> this wave's pieces of unit u+1 have landed (u+2, u+3 stay in flight), its reads are done
//     s_barrier                      -> unit u+1 visible to everyone, slot (u-0)&3 free for the DMA issued in unit u+1
//     ds_read B(u+1), A(u+1,h0) | MFMA (u,h1)
// One barrier per 32 MFMAs per wave; every DMA has three units (about 3 x 1024 matrix-pipe cycles per SIMD) to land.  Units past
// the end of K are issued as out-of-range pieces so that the vmcnt arithmetic stays uniform.
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
#define V3_SLOTS 4
#define V3_LDS (V3_SLOTS * V3_UNIT)        // 128 KiB = the epilogue's [256][256] bf16 image

// G[(r>>2)&3] of the header: {0,2,3,1}
__device__ __forceinline__ int v3_swz(int r) {
  const int q = (r >> 2) & 3;
  return (((q ^ (q >> 1)) & 1) << 1) | (q >> 1);
}

// EPI 1 (data gradient only): the result is the gradient of a BatchNorm(+ReLU) output -- it is (accumulated into dst when GF_ACCUM,
// then) masked with that BatchNorm's ReLU decision (sign bytes F.ep_mask [pixel][NC/4], or recomputed from the raw tensor F.ep_x
// with F.ep_fcoef when GF_EPI_RELU), stored, and the partial sums (sum g, sum g*(x - mean)) of every 128-row chunk go to
// F.ep_part [F.ep_chunk0 + row / 128][2][NC] -- the layout the 128-row kernel's epilogue emits (conv_bf16.hip EPI 1), so the
// BatchNorm backward of the wide residual stages needs no reduction pass of its own over the gradient.
template <bool DGRAD, int DBG = 0, bool STAGGER = false, int PRIO = 2, int EPI = 0>
__global__ __launch_bounds__(512, 2) void conv_gather_bf16_v3_kernel(const __bf16* __restrict__ src, const __bf16* __restrict__ wm,
                                                                     __bf16* __restrict__ dst, GatherGeom g, int tiles_n,
                                                                     GatherFuse F) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TM = 8, TN = 4;                 // wave tile 128 pixels x 64 channels of 16x16 MFMA tiles
  unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};  // DBG 4 only: s_memtime at entry / loop start / loop end / kernel end, s_memrealtime at entry / end
  auto cstamp = [&](int i) {
    if constexpr (DBG == 4) {
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[i])::"memory");
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  unsigned long long* F_dbg = DBG == 4 ? (unsigned long long*)g.stat_shift : nullptr;
  cstamp(0);
  if constexpr (DBG == 4) asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[4])::"memory");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;
  const int lid = edrl_xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const long m0 = (long)tile_m * V3_BM;
  const int n0 = tile_n * V3_BN;

  // ---- DMA addressing: thread -> row (tid >> 2) + 128 j of both operand units, LDS slot tid & 3 = k chunk (tid & 3) ^ G[row]
  const int srow = tid >> 2;                                   // 0..127
  const int kc8 = (((tid & 3) ^ v3_swz(srow)) * 8);
  const int ohw = g.OHs * g.OWs;
  const int n_first = (int)(m0 / ohw);
  int pb[2], hw[2];
  // 1x1 / stride 1 / pad 0 (forward and data gradient alike): source pixel = output row, no (n, oh, ow) decode -- the two 64-bit
  // divisions per thread are a fifth of the prologue of a short-K tile
  const bool lin = g.KH == 1 && g.KW == 1 && g.pad == 0 && g.stride == 1 && g.step == 1 && g.SH == g.OHs && g.SW == g.OWs;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const long m = m0 + srow + 128 * j;
    if (lin) {
      pb[j] = m < g.M ? (int)(m - (long)n_first * ohw) : -1;
      hw[j] = (16384 << 16) | 16384;
    } else if (m < g.M) {
      const int n = (int)(m / ohw);
      const int rem = (int)(m - (long)n * ohw);
      const int ii = rem / g.OWs, jj = rem - ii * g.OWs;
      const int oh = g.h0 + ii * g.step, ow = g.w0 + jj * g.step;
      int rh, rw;
      if (DGRAD) { rh = oh + g.pad; rw = ow + g.pad; }
      else       { rh = oh * g.stride - g.pad; rw = ow * g.stride - g.pad; }
      pb[j] = (n - n_first) * g.SH * g.SW;
      hw[j] = ((rh + 16384) << 16) | (rw + 16384);
    } else { pb[j] = -1; hw[j] = 0; }
  }
  constexpr unsigned OOB = 0x80000000u;
  long mlast = m0 + V3_BM; if (mlast > g.M) mlast = g.M;
  const int n_last = (int)((mlast - 1) / ohw);
  const unsigned a_bytes = (unsigned)(((long)(n_last - n_first + 1) * g.SH * g.SW - 1) * g.ld_src * 2 + (long)g.SC * 2);
  const v3_i32x4 rs_a = v3_make_srd(src + (long)n_first * g.SH * g.SW * g.ld_src, a_bytes);
  const v3_i32x4 rs_b = v3_make_srd(wm, (unsigned)((long)g.NC * g.Kfull * 2));
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)(lds_ptr_t)smem) + (unsigned)wave * 1024u;
  // weight rows n0 + srow + 128 j: NC % 256 == 0 (host-checked), so every row exists; the 128 j part rides in the scalar offset
  const unsigned wrow0 = (unsigned)(n0 + srow) * (unsigned)g.Kfull * 2u;
  const int wj = 128 * g.Kfull * 2;
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
  // pieces of one unit: 2 of the pixel operand, 2 of the weights; `slot` is wave-uniform
  auto issueA = [&](int slot) {
    if constexpr (DBG == 2) return;
    const unsigned base = lds0 + (unsigned)slot * V3_UNIT;
    v3_dma16(base, DBG == 1 ? (aoff[0] & 0xfff0u) : aoff[0], rs_a, 0);
    v3_dma16(base + 8192, DBG == 1 ? (aoff[1] & 0xfff0u) : aoff[1], rs_a, 0);
  };
  auto issueB = [&](int slot) {
    if constexpr (DBG == 2) return;
    const unsigned base = lds0 + (unsigned)slot * V3_UNIT + V3_ABYTES;
    v3_dma16(base, DBG == 1 ? (boff & 0xfff0u) : boff, rs_b, 0);
    v3_dma16(base + 8192, DBG == 1 ? (boff & 0xfff0u) : boff, rs_b, DBG == 1 ? 0 : wj);
  };

  auto issueA1 = [&](int slot, int j) {
    if constexpr (DBG == 2) return;
    v3_dma16(lds0 + (unsigned)slot * V3_UNIT + (unsigned)j * 8192u, DBG == 1 ? (aoff[j] & 0xfff0u) : aoff[j], rs_a, 0);
  };
  auto issueB1 = [&](int slot, int j) {
    if constexpr (DBG == 2) return;
    v3_dma16(lds0 + (unsigned)slot * V3_UNIT + V3_ABYTES + (unsigned)j * 8192u, DBG == 1 ? (boff & 0xfff0u) : boff, rs_b,
             (DBG == 1 || j == 0) ? 0 : wj);
  };

  // ---- fragment addressing (bytes inside a unit): row fr (+16 i), k chunk fq at slot fq ^ G[fr]
  const int fr = lane & 15, fq = lane >> 4;
  const int a_rd = (wm0 + fr) * 64 + ((fq ^ v3_swz(fr)) << 4);
  const int b_rd = V3_ABYTES + (wn0 + fr) * 64 + ((fq ^ v3_swz(fr)) << 4);

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // register sets: two halves of the pixel fragments (4 tiles each), weight fragments of the current and the next unit
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
  // 8 MFMAs: pixel tiles 2q, 2q+1 of half MH against the 4 weight tiles
  auto mma8 = [&](auto MH_, auto Q_, bf16x8 (&af)[4], bf16x8 (&bf)[TN]) {
    constexpr int MH = decltype(MH_)::value, Q = decltype(Q_)::value;
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int j = 2 * Q; j < 2 * Q + 2; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i)
        acc[i][MH * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[i], af[j], acc[i][MH * 4 + j], 0, 0, 0);
    if constexpr (PRIO == 1) __builtin_amdgcn_s_setprio(0);
  };
  using H0 = std::integral_constant<int, 0>;
  using H1 = std::integral_constant<int, 1>;
  unsigned long long seg[5] = {0, 0, 0, 0, 0};      // DBG 3 only: cycles in {reads + first half, wait, barrier, second half, -}
  auto stamp = [&]() -> unsigned long long {
    if constexpr (DBG != 3) return 0ull;
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
  };
  // The vector-memory path of a CU moves ~64 B/clk: the 32 KiB of one unit take >= 512 cycles of it, and a wave whose DMA
  // instruction finds the queue full stalls in issue -- with all 8 waves issuing their 4 pieces at one program point the last
  // ones waited ~600 cycles per unit with no MFMA of theirs in the pipe (in-kernel stamps, scripts/dbg/v3_stamps.py).  So the
  // pieces are spread over the unit, ONE in front of every group of 8 MFMAs (pixel-operand pieces in the first half, weight
  // pieces -- whose slot is free as soon as the unit's barrier has passed -- in the second half), and with STAGGER the two
  // waves that share a SIMD (w, w+4) alternate: one issues its piece before its 8 MFMAs, the other after.
  const bool dma_first = STAGGER ? wave < 4 : true;
  // one unit: `cur` weights are multiplied, `nxt` receives the next unit's
  auto unit = [&](int u, bf16x8 (&bcur)[TN], bf16x8 (&bnxt)[TN]) {
    const int slot = u & 3, nslot = (u + 3) & 3;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = stamp();
    rdA(slot, 1, an);
    if (dma_first) issueA1(nslot, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H0{}, H0{}, ac, bcur);
    __builtin_amdgcn_sched_barrier(0);
    if (!dma_first) issueA1(nslot, 0); else issueA1(nslot, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H0{}, H1{}, ac, bcur);
    __builtin_amdgcn_sched_barrier(0);
    if (!dma_first) issueA1(nslot, 1);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = stamp();
    // this wave's pieces of unit u+1 have landed: of the later ones, unit u+2 (4) and the pixel pieces of unit u+3 (2) may fly
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    const unsigned long long t2 = stamp();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t3 = stamp();
    rdB((u + 1) & 3, bnxt);
    rdA((u + 1) & 3, 0, ac);
    if (dma_first) issueB1(nslot, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H1{}, H0{}, an, bcur);
    __builtin_amdgcn_sched_barrier(0);
    if (!dma_first) issueB1(nslot, 0); else issueB1(nslot, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H1{}, H1{}, an, bcur);
    __builtin_amdgcn_sched_barrier(0);
    if (!dma_first) issueB1(nslot, 1);
    advance();
    const unsigned long long t4 = stamp();
    if constexpr (DBG == 3) { seg[0] += t1 - t0; seg[1] += t2 - t1; seg[2] += t3 - t2; seg[3] += t4 - t3; }
  };

  // Static priority for the second-dispatched half (MI355X_MICROARCH.md "Two waves per SIMD", item 4) instead of s_setprio flips
  // around every MFMA group (PRIO 1): measured +2..5 % on the 3x3 layers (1034 -> 1085, 1110 -> 1133 TFLOP/s), no flips +2..4 %.
  if constexpr (PRIO == 2) { if (wave >= 4) __builtin_amdgcn_s_setprio(1); }
  if (KU > 0) {
    retap();
#pragma unroll
    for (int u = 0; u < 3; ++u) { issueA(u); issueB(u); advance(); }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // unit 0 landed (this wave's 4 pieces), units 1 and 2 still in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    rdB(0, bc);
    rdA(0, 0, ac);
    cstamp(1);
    int u = 0;
    for (; u + 1 < KU; u += 2) {
      unit(u, bc, bn);
      unit(u + 1, bn, bc);
    }
    if (u < KU) unit(u, bc, bn);
    // the pipeline's tail pieces (zeros into consumed slots) must have landed before the epilogue reuses the LDS
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    cstamp(2);
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (DBG == 3) {
    if (lane == 0 && g.stat_part) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(g.stat_part) + ((long)blockIdx.x * 8 + wave) * 8;
#pragma unroll
      for (int i = 0; i < 5; ++i) o[i] = seg[i];
    }
    return;
  }
  __syncthreads();

  // ---- BatchNorm chunk partials from the fp32 accumulators (forward, GF_STATS): this wave owns one 128-row chunk x 64 channels
  if (!DGRAD && (g.flags & GF_STATS)) {
    const long crow0 = m0 + wm0;
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
        const int n = n0 + wn0 + i * 16 + 4 * fq;
        if (fr == 0 && n < g.NC) {
          *reinterpret_cast<f32x4*>(pp + n) = s1;
          *reinterpret_cast<f32x4*>(pp + g.NC + n) = s2;
          *reinterpret_cast<f32x4*>(pp + 2 * (long)g.NC + n) = kk;
        }
      }
    }
  }

  // ---- bf16 result -> LDS image [256 pixels][256 channels] (512-byte rows, 16-byte chunk c of row r at slot c ^ (r & 15))
  {
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
  }
  __syncthreads();
  if constexpr (EPI != 0) {
    const bool accum = g.flags & GF_ACCUM;
    const int rr = tid >> 5, c = tid & 31;
    const int n = n0 + c * 8;                       // (NC % 256 == 0: every column of the tile exists)
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
    for (int h = 0; h < 2; ++h) {                   // chunk h of the tile = rows 128 h .. 128 h + 127
      long pixs[8];
      bf16x8 opre[8], xpre[8];
      unsigned mb[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {                 // all global operands of the chunk requested up front
        const long m = m0 + (h * 8 + i) * 16 + rr;
        long pix = m;
        if (g.step > 1 && m < g.M) {
          const int nn = (int)(m / ohw);
          const int rem = (int)(m - (long)nn * ohw);
          const int ii = rem / g.OWs, jj = rem - ii * g.OWs;
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
          const int row = (h * 8 + i) * 16 + rr;
          const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + row * 512 + ((c ^ rr) << 4));
          bf16x8 ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float vf = (float)v[e];
            if (accum) vf += (float)opre[i][e];
            const float xe = (float)xpre[i][e];
            const bool keep = use_mask ? ((mb[i] >> ((e & 3) + 8 * (e >> 2))) & 1u) != 0u :      // two sign bytes, 4 channels each
                                        (use_relu ? __builtin_fmaf(xe, esc[e], esh[e]) > 0.f : true);
            vf = keep ? vf : 0.f;
            s0[h][e] += vf;
            s1[h][e] = __builtin_fmaf(vf, xe - em[e], s1[h][e]);
            ov[e] = (__bf16)vf;
          }
          *reinterpret_cast<bf16x8*>(dst + pixs[i] * g.ld_dst + n) = ov;
        }
      }
    }
    __syncthreads();                                // every wave is done with the tile image: its LDS becomes the reduction scratch
    float* red = reinterpret_cast<float*>(smem);    // [chunk][plane][wave][256 channels]
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
    for (int o = tid; o < 1024; o += 512) {         // ordered sum over the 8 waves: deterministic
      const int h = o >> 9, pl = (o >> 8) & 1, ch = o & 255;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) t += red[((h * 2 + pl) * 8 + w) * 256 + ch];
      if (m0 + h * 128 < g.M) F.ep_part[((long)F.ep_chunk0 + 2 * tile_m + h) * 2 * g.NC + (long)pl * g.NC + n0 + ch] = t;
    }
  } else {
    const bool accum = g.flags & GF_ACCUM;
    const int rr = tid >> 5, c = tid & 31;
    const int n = n0 + c * 8;
    long pixs[16];
    bf16x8 opre[16];
    // destination rows first; when accumulating, all 16 read-modify-write operands of the thread are requested up front
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const long m = m0 + it * 16 + rr;
      long pix = m;
      if (DGRAD && g.step > 1 && m < g.M) {
        const int nn = (int)(m / ohw);
        const int rem = (int)(m - (long)nn * ohw);
        const int ii = rem / g.OWs, jj = rem - ii * g.OWs;
        pix = ((long)nn * g.OH + g.h0 + ii * g.step) * g.OW + g.w0 + jj * g.step;
      }
      pixs[it] = (m < g.M && n < g.NC) ? pix : -1;
      if (accum && pixs[it] >= 0) opre[it] = *reinterpret_cast<const bf16x8*>(dst + pix * g.ld_dst + n);
    }
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      if (pixs[it] >= 0) {
        const int row = it * 16 + rr;
        bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + row * 512 + ((c ^ rr) << 4));
        if (accum) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (__bf16)((float)v[e] + (float)opre[it][e]);
        }
        *reinterpret_cast<bf16x8*>(dst + pixs[it] * g.ld_dst + n) = v;
      }
    }
  }
  if constexpr (DBG == 4) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    cstamp(3);
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts[5])::"memory");
    if (lane == 0 && F_dbg) {
      unsigned long long* o = F_dbg + ((long)blockIdx.x * 8 + wave) * 8;
#pragma unroll
      for (int q = 0; q < 6; ++q) o[q] = ts[q];
    }
  }
}

bool gather_bf16_v3_ok(const GatherGeom& g, bool dgrad) {
  const int mode = edrl_cfg().bf16_v3;                 // EDRL_BF16_V3: 0 off, 1 auto (default), 2 force wherever the geometry allows
  if (mode == 0) return false;
  const long ohw = (long)g.OHs * g.OWs;
  const bool can = (g.SC % V3_BK == 0) && (g.NC % V3_BN == 0) && (g.ld_dst % 8 == 0) && (g.ld_src % 8 == 0) && ohw > 0 && g.M > 0 &&
                   (V3_BM / ohw + 2) * g.SH * g.SW * g.ld_src * 2 < (1L << 31) && (long)g.NC * g.Kfull * 2 < (1L << 31) &&
                   (g.Ktot % V3_BK == 0);
  if (!can) return false;
  if (mode == 2) return true;
  // Measured per layer against the 128-row kernel (scripts/v3_layer_bench.py, profiles/r04_v3_layers_bf16_2112img.txt): in its
  // persistent form (conv_bf16_v3p.hip) the forward wins from K = 256 up (`l3 1x1 256->1024` 0.462 -> 0.425 ms; round 3's
  // one-tile-per-workgroup form won from K = 512 only and lost at K = 256: nothing hid its prologue / epilogue behind 8 K units);
  // K = 128 still loses (`l2 1x1 128->512` 0.505 vs 0.680 ms).  The data gradient wins at every K that occurs (>= 128), its
  // 128-row counterpart pays more for the scattered / accumulating epilogue.  Both need enough rows to fill the chip once.
  const int kmin_f = edrl_cfg().v3_fwd_kmin;
  return g.M >= 256 * 64 && g.Ktot >= (dgrad ? 128 : kmin_f);
}

bool gather_bf16_v3_epi_ok(const GatherGeom& g, const GatherFuse& F) {
  return F.ep_x && F.ep_fcoef && F.ep_part && (F.ld_ep % 8 == 0) && !((uintptr_t)F.ep_x & 15) && !((uintptr_t)F.ep_mask & 1) &&
         (g.NC % 8 == 0);
}

int launch_gather_bf16_v3(const void* src, const void* wm, void* dst, const GatherGeom& g, bool dgrad, hipStream_t st,
                          const GatherFuse* fuse) {
  const int tiles_m = edrl_cdiv(g.M, V3_BM), tiles_n = edrl_cdiv(g.NC, V3_BN);
  const long nblk = (long)tiles_m * tiles_n;
  if (nblk <= 0) return 0;
  if (nblk > 0x7fffffffL) return EDRL_EINVAL;
  if (((uintptr_t)src & 15) || ((uintptr_t)wm & 15) || ((uintptr_t)dst & 15)) return EDRL_EINVAL;
  static bool attr_set[3] = {false, false, false};
  GatherFuse F;
  memset(&F, 0, sizeof(F));
#ifdef EDRL_DIAG
  const bool diag = edrl_cfg().diag_v3 != 0;
#else
  const bool diag = false;
#endif
  if (!diag && edrl_cfg().v3_stagger != 1 && gather_bf16_v3p_ok(g, fuse && fuse->ep_x)) {      // the persistent form (conv_bf16_v3p.hip)
    if (fuse && fuse->ep_x && (!dgrad || !gather_bf16_v3_epi_ok(g, *fuse))) return EDRL_EINVAL;
    return launch_gather_bf16_v3p(src, wm, dst, g, dgrad, st, fuse);
  }

  if (fuse && fuse->ep_x) {      // data gradient with the BatchNorm-backward epilogue (mask + partial sums)
    if (!dgrad || !gather_bf16_v3_epi_ok(g, *fuse)) return EDRL_EINVAL;
    auto ke = conv_gather_bf16_v3_kernel<true, 0, false, 2, 1>;
    if (!attr_set[2]) { (void)hipFuncSetAttribute((const void*)ke, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr_set[2] = true; }
    hipLaunchKernelGGL(ke, dim3((unsigned)nblk), dim3(512), V3_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, *fuse);
    EDRL_LAUNCH_CHECK();
    return 0;
  }
  // EDRL_V3_STAGGER=1: the staggered variant (A/B switch; measured 0-4 % slower than the plain order, see the kernel comment)
  if (edrl_cfg().v3_stagger == 1) {
    auto ks = dgrad ? conv_gather_bf16_v3_kernel<true, 0, true> : conv_gather_bf16_v3_kernel<false, 0, true>;
    (void)hipFuncSetAttribute((const void*)ks, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS);
    hipLaunchKernelGGL(ks, dim3((unsigned)nblk), dim3(512), V3_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, F);
    EDRL_LAUNCH_CHECK();
    return 0;
  }
  if (dgrad) {
    auto kern = conv_gather_bf16_v3_kernel<true>;
    if (!attr_set[1]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr_set[1] = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), V3_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, F);
  } else {
#ifdef EDRL_DIAG
    // Diagnostic builds (wrong or absent outputs by construction): compiled only into libedrl_hip_diag.so (make diag), never into
    // the shipped library, so no environment can select them in production.
    const int dbg = edrl_cfg().diag_v3;
    if (dbg >= 1 && dbg <= 4) {     // diagnostic builds: 1 cache-resident loads, 2 no DMA, 3 / 4 in-kernel stamps
      auto kd = dbg == 1 ? conv_gather_bf16_v3_kernel<false, 1> : (dbg == 2 ? conv_gather_bf16_v3_kernel<false, 2> : (dbg == 3 ? conv_gather_bf16_v3_kernel<false, 3> : conv_gather_bf16_v3_kernel<false, 4>));
      (void)hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS);
      GatherGeom gd = g;
      if (dbg == 4) { gd.stat_shift = (const float*)g.stat_part; gd.flags &= ~GF_STATS; }     // stamps go to the caller's partials buffer
      hipLaunchKernelGGL(kd, dim3((unsigned)nblk), dim3(512), V3_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, gd, tiles_n, F);
      EDRL_LAUNCH_CHECK();
      return 0;
    }
#endif
    auto kern = conv_gather_bf16_v3_kernel<false>;
    if (!attr_set[0]) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, V3_LDS); attr_set[0] = true; }
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), V3_LDS, st, (const __bf16*)src, (const __bf16*)wm, (__bf16*)dst, g, tiles_n, F);
  }
  EDRL_LAUNCH_CHECK();
  return 0;
}
