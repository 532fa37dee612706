// 3x3 / stride 1 / pad 1 convolution between two 64-channel bf16 tensors: the middle layer of the first residual stage's bottleneck
// blocks (ResNet-50 layer1.*.conv2, forward and data gradient; SURVEY.md section 8a rows E1/E2 at C2/C4).  With N = 64 output
// channels the 128-row implicit-GEMM kernel re-stages a 64 x 32 weight tile per K tile and workgroup and reaches ~490 TFLOP/s of
// the 2.5 PFLOP/s peak (1.0 ms per 2112 images for 0.49 TFLOP); this layer is 9 % of the C2 step's conv time.
//
// Structure: the whole weight tensor (64 x 576 bf16 = 72 KiB) sits in LDS for the lifetime of a persistent workgroup, already in
// MFMA-fragment order (one ds_read_b128 per fragment, lane-linear: conflict-free).  The pixel operand never touches LDS: one wave
// covers ALL 64 output channels of its pixels, so nobody else needs its pixel fragments -- each lane loads its 16 bytes (8 channels of
// one kernel row of one pixel) straight into the fragment register with a buffer load (out-of-image rows and rows past the end are
// out-of-range offsets: the hardware returns zeros), two load groups ahead; the left / right taps of a kernel row are the centre
// fragment shifted by one pixel lane (DPP), so every input pixel is fetched 3 times, not 9.  No barrier after the weight fill: the 8 waves of a
// workgroup run independently over 128-pixel chunks (2 blocks of 64 pixels = 4 MFMA column tiles; 18 K steps x 16 MFMAs per block).
//
// v_mfma_f32_16x16x32_bf16 with the WEIGHT fragment first: a lane ends with 4 consecutive channels of one pixel per tile; lanes g and
// g^1 (lane ^ 16) swap two tiles so that each holds 8 consecutive channels twice -> 16-byte bf16 stores (and 16-byte epilogue operand
// loads).  A lane owns the same 16 channels for every pixel it ever sees, so the per-chunk partial sums -- BatchNorm statistics
// (sum (y-K), sum (y-K)^2, K = the chunk's first row) in the forward, (sum g, sum g*x) of the masked gradient in the data gradient --
// are 32 registers accumulated over the chunk and reduced over the 16 pixel lanes once per 128 pixels, from the unrounded fp32
// accumulators (same contracts as conv_bf16.hip's epilogues).
#include "edrl_common.h"
#include "edrl_config.h"
#include <stdlib.h>
#include <string.h>
#include "conv_bf16_v3.h"

typedef __bf16 c64_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int c64_u32x4 __attribute__((ext_vector_type(4)));

#define C64_WBYTES (18 * 4 * 64 * 16)  // 72 KiB: [K step][channel tile][lane] x 16 bytes
#define C64_LDS (C64_WBYTES + 256)     // + the 64 batch means of the EPI 1 epilogue

struct C64Geom {
  int M, H, W, nchunks;
  unsigned m_w, m_h;       // magic multipliers: x / d == (x * m) >> k for x < 2^24 (host-checked)
  int k_w, k_h;
};

// EPI 0: bf16 store (+ BatchNorm chunk partials [chunk][3][64] when stat_part != NULL)
// EPI 1: the result is the gradient of a BatchNorm+ReLU output: masked with the sign bytes ep_mask [M][16], stored as bf16,
//        (sum g, sum g*(x - mean)) with x = ep_x [M][64] (raw conv output of that BatchNorm, mean = ep_mean [64]) -> ep_part [chunk][2][64]
template <int EPI, int DBG = 0>
__global__ __launch_bounds__(512, 2) void conv3x3_c64_bf16_kernel(const __bf16* __restrict__ src, const __bf16* __restrict__ w,
                                                                  __bf16* __restrict__ dst, C64Geom g, int flip,
                                                                  float* __restrict__ stat_part, const __bf16* __restrict__ ep_x,
                                                                  const unsigned char* __restrict__ ep_mask,
                                                                  float* __restrict__ ep_part, const float* __restrict__ ep_mean) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // EPI 1: the batch mean of ep_x's BatchNorm [64] behind the weights (plane 1 of the partials is sum g*(x - mean))
  if constexpr (EPI == 1) { if (tid < 64) reinterpret_cast<float*>(smem + C64_WBYTES)[tid] = ep_mean[tid]; }
  // ---- weights -> LDS in fragment order: fragment (K step ks = 2*tap + half, channel tile ct), lane l = (row l&15 = channel
  // 16 ct + (l&15), k chunk l>>4 = input channels 32 half + 8 (l>>4) .. +7).  flip (data gradient, w = [Ci][3][3][Co]): tap 8 - t.
  for (int idx = tid; idx < 18 * 4 * 64; idx += 512) {
    const int l = idx & 63, ct = (idx >> 6) & 3, ks = idx >> 8;
    const int n = 16 * ct + (l & 15);
    const int tap = ks >> 1, tsrc = flip ? 8 - tap : tap;
    *reinterpret_cast<c64_bf16x8*>(smem + idx * 16) =
        *reinterpret_cast<const c64_bf16x8*>(w + ((long)(n * 9 + tsrc)) * 64 + 32 * (ks & 1) + 8 * (l >> 4));
  }
  __syncthreads();

  const int pl = lane & 15, gq = lane >> 4;
  const bool even = (gq & 1) == 0;
  const int cb0 = even ? 4 * gq : 16 + 4 * (gq - 1);      // this lane's two runs of 8 consecutive output channels: cb0, cb0 + 32
  constexpr unsigned OOB = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)((long)g.M * 128), 0x00020000);
  const int wrow = g.W * 128;                               // bytes per image row
  const unsigned char* wl = smem + lane * 16;

  // ---- this lane's 4 pixels of a block (one per column tile): byte offset of the centre tap, 9-bit validity of the taps
  auto decode = [&](int m0, unsigned (&off0)[4], int (&tmask)[4]) {
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int m = m0 + 16 * pt + pl;
      const unsigned r = (unsigned)(((unsigned long long)(unsigned)m * g.m_w) >> g.k_w);       // m / W
      const int ow = m - (int)r * g.W;
      const unsigned q = (unsigned)(((unsigned long long)r * g.m_h) >> g.k_h);                   // (m / W) / H
      const int oh = (int)r - (int)q * g.H;
      int tm = m < g.M ? 0x1ff : 0;
      if (oh == 0) tm &= ~0x007;
      if (oh == g.H - 1) tm &= ~0x1c0;
      if (ow == 0) tm &= ~0x049;
      if (ow == g.W - 1) tm &= ~0x124;
      tmask[pt] = tm;
      off0[pt] = (unsigned)m * 128u + (unsigned)gq * 16u;
    }
  };
  // One load per (kernel row kh, channel half h) and pixel tile: the three taps of a kernel row read pixels m-1, m, m+1, i.e. the
  // centre fragment shifted by one pixel LANE (DPP row_shr / row_shl inside the 16-lane pixel rows of the fragment; the lane at the
  // end of a row takes the neighbouring tile's end lane, tile 0 / 3 the two edge pixels loaded by `e`) -- a third of the loads and of
  // the L2 traffic of loading every tap.  Shifted values that belong to another image row or lie outside the image are zeroed by
  // the tap mask (the centre fragment's out-of-image rows are zero from the range check already).
  auto load = [&](int grp, int m0, const unsigned (&off0)[4], const int (&tmask)[4], c64_u32x4 (&c)[4], c64_u32x4& e) {
    const int kh = grp >> 1;
    const int toff = (kh - 1) * wrow + (grp & 1) * 64;
    if constexpr (DBG == 2) {
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) c[pt] = c64_u32x4{off0[pt], 1u, 2u, 3u};
      e = c64_u32x4{0u, 0u, 0u, 0u};
      return;
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const unsigned vo = ((tmask[pt] >> (3 * kh + 1)) & 1) ? off0[pt] + (unsigned)toff : OOB;
      c[pt] = __builtin_bit_cast(c64_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)vo, 0, 0));
    }
    // edge pixels: lane 15 of each row <- pixel m0 - 1, lane 0 <- pixel m0 + 64 (offsets outside the tensor wrap out of range)
    const unsigned eo = pl == 15 ? (unsigned)(m0 - 1) * 128u + (unsigned)gq * 16u + (unsigned)toff
                      : pl == 0 ? (unsigned)(m0 + 64) * 128u + (unsigned)gq * 16u + (unsigned)toff : OOB;
    e = __builtin_bit_cast(c64_u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)eo, 0, 0));
  };
  // lane i <- src[i-1] of its 16-lane row, lane 0 <- lane 15 of `lo` (the tile to the left)
  auto shr1 = [&](c64_u32x4 src, c64_u32x4 lo, bool keep) {
    c64_u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int t = __builtin_amdgcn_update_dpp(0, (int)lo[d], 0x121, 0xf, 0xf, false);          // row_ror:1
      const int v = __builtin_amdgcn_update_dpp(t, (int)src[d], 0x111, 0xf, 0xf, false);         // row_shr:1 (lane 0 keeps t)
      o[d] = keep ? (unsigned)v : 0u;
    }
    return o;
  };
  auto shl1 = [&](c64_u32x4 src, c64_u32x4 hi, bool keep) {
    c64_u32x4 o;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int t = __builtin_amdgcn_update_dpp(0, (int)hi[d], 0x12f, 0xf, 0xf, false);          // row_ror:15: lane 15 <- lane 0
      const int v = __builtin_amdgcn_update_dpp(t, (int)src[d], 0x101, 0xf, 0xf, false);         // row_shl:1 (lane 15 keeps t)
      o[d] = keep ? (unsigned)v : 0u;
    }
    return o;
  };
  // row-wise running sums by DPP (row_shr 8 / 4 / 2 / 1, lanes shifted in from outside the row read 0): lane 15 of each 16-lane row
  // ends with the row's total -- vector ALU beside the MFMAs instead of LDS-crossbar shuffles
  auto rowsum = [&](float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
    return x;
  };

  // Blocks of 64 pixels, two per 128-pixel chunk; chunk c of this wave = c0 + k * cs.  Software pipeline over blocks: the first two
  // load groups of block i+1 are issued between the MFMAs and the epilogue of block i.
  const int c0 = blockIdx.x * 8 + wave, cs = gridDim.x * 8;
  auto block_m0 = [&](int it) {          // first pixel of block `it` of this wave, -1: none
    const long chunk = (long)c0 + (long)(it >> 1) * cs;
    if (chunk >= g.nchunks) return -1;
    const long m0 = chunk * 128 + (it & 1) * 64;
    return m0 < g.M ? (int)m0 : -1;
  };
  c64_u32x4 cf[3][4], ef[3];
  unsigned off0[4];
  int tmask[4];
  int m0 = block_m0(0);
  if (m0 >= 0) {
    decode(m0, off0, tmask);
    load(0, m0, off0, tmask, cf[0], ef[0]);
    load(1, m0, off0, tmask, cf[1], ef[1]);
  }
  f32x4 sa[4], sb[4], kk[4];               // per-chunk partial sums of this lane's 16 channels (quads: cb0, cb0+4, cb0+32, cb0+36)
#pragma unroll 1
  for (int it = 0;; ++it) {
    const long chunk = (long)c0 + (long)(it >> 1) * cs;
    if (chunk >= g.nchunks) break;
    if ((it & 1) == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) sa[q] = sb[q] = kk[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int m0n = block_m0(it + 1);
    if (m0 >= 0) {
      f32x4 acc[4][4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int grp = 0; grp < 6; ++grp) {
        if (grp + 2 < 6) load(grp + 2, m0, off0, tmask, cf[(grp + 2) % 3], ef[(grp + 2) % 3]);
        const int kh = grp >> 1, h = grp & 1;
        c64_u32x4 (&c)[4] = cf[grp % 3];
        const c64_u32x4 e = ef[grp % 3];
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int tap = 3 * kh + kw;
          c64_bf16x8 wf[4];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) wf[ct] = *reinterpret_cast<const c64_bf16x8*>(wl + ((2 * tap + h) * 4 + ct) * 1024);
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) {
            const bool keep = (tmask[pt] >> tap) & 1;
            c64_u32x4 a;
            if (kw == 1 || DBG == 1) a = c[pt];
            else if (kw == 0) a = shr1(c[pt], pt > 0 ? c[pt - 1] : e, keep);
            else a = shl1(c[pt], pt < 3 ? c[pt + 1] : e, keep);
            const c64_bf16x8 ab = __builtin_bit_cast(c64_bf16x8, a);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
              acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct], ab, acc[ct][pt], 0, 0, 0);
          }
        }
      }
      // ---- the next block's first loads fly during this block's epilogue
      const int m0c = m0;
      if (m0n >= 0) {
        decode(m0n, off0, tmask);
        load(0, m0n, off0, tmask, cf[0], ef[0]);
        load(1, m0n, off0, tmask, cf[1], ef[1]);
      }
      // ---- epilogue of the block
#pragma unroll
      for (int pt = 0; pt < 4; ++pt) {
        f32x4 v[4];          // quads of channels cb0, cb0+4, cb0+32, cb0+36 of pixel m
        // v_permlane16_swap: the odd 16-lane rows of the first register trade places with the even rows of the second -- an even
        // row keeps its tile 0 and receives its neighbour row's tile 0 (the next 4 channels), an odd row receives the neighbour's tile 1
        // and keeps its own: 8 consecutive channels per lane, one instruction per register pair
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const auto p0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][pt][e]), __float_as_uint(acc[1][pt][e]), false, false);
          const auto p1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2][pt][e]), __float_as_uint(acc[3][pt][e]), false, false);
          v[0][e] = __uint_as_float(p0[0]); v[1][e] = __uint_as_float(p0[1]);
          v[2][e] = __uint_as_float(p1[0]); v[3][e] = __uint_as_float(p1[1]);
        }
        const int m = m0c + 16 * pt + pl;
        const bool ok = m < g.M;
        if constexpr (EPI == 0) {
          if (stat_part) {
            if ((it & 1) == 0 && pt == 0) {
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) kk[q][e] = __shfl(v[q][e], lane & 48, 64);      // the chunk's first row
            }
            if (ok) {
#pragma unroll
              for (int q = 0; q < 4; ++q) { const f32x4 d = v[q] - kk[q]; sa[q] += d; sb[q] = __builtin_elementwise_fma(d, d, sb[q]); }
            }
          }
        } else {
          if (ok) {
            f32x4 em[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
              em[q] = *reinterpret_cast<const f32x4*>(smem + C64_WBYTES + 4 * (cb0 + 4 * (q & 1) + 32 * (q >> 1)));
            const c64_bf16x8 x0 = *reinterpret_cast<const c64_bf16x8*>(ep_x + (long)m * 64 + cb0);
            const c64_bf16x8 x1 = *reinterpret_cast<const c64_bf16x8*>(ep_x + (long)m * 64 + cb0 + 32);
            const unsigned mb0 = *reinterpret_cast<const unsigned short*>(ep_mask + (long)m * 16 + (cb0 >> 2));
            const unsigned mb1 = *reinterpret_cast<const unsigned short*>(ep_mask + (long)m * 16 + (cb0 >> 2) + 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const unsigned mb = ((q < 2 ? mb0 : mb1) >> (8 * (q & 1)));
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const float ve = (mb >> e) & 1 ? v[q][e] : 0.f;
                v[q][e] = ve;
                const float xe = (float)(q < 2 ? x0 : x1)[4 * (q & 1) + e] - em[q][e];
                sa[q][e] += ve;
                sb[q][e] = __builtin_fmaf(ve, xe, sb[q][e]);
              }
            }
          }
        }
        if (ok) {
          c64_bf16x8 o0, o1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            o0[e] = (__bf16)v[0][e]; o0[4 + e] = (__bf16)v[1][e];
            o1[e] = (__bf16)v[2][e]; o1[4 + e] = (__bf16)v[3][e];
          }
          *reinterpret_cast<c64_bf16x8*>(dst + (long)m * 64 + cb0) = o0;
          *reinterpret_cast<c64_bf16x8*>(dst + (long)m * 64 + cb0 + 32) = o1;
        }
      }
    } else if (m0n >= 0) {
      decode(m0n, off0, tmask);
      load(0, m0n, off0, tmask, cf[0], ef[0]);
      load(1, m0n, off0, tmask, cf[1], ef[1]);
    }
    m0 = m0n;
    // ---- chunk partials: sum over the 16 pixel lanes, one lane per channel run writes
    float* part = EPI == 0 ? stat_part : ep_part;
    if ((it & 1) && part) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) { sa[q][e] = rowsum(sa[q][e]); sb[q][e] = rowsum(sb[q][e]); }
      if (pl == 15) {
        float* pp = part + chunk * (EPI == 0 ? 3 : 2) * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = cb0 + 4 * (q & 1) + 32 * (q >> 1);
          *reinterpret_cast<f32x4*>(pp + c) = sa[q];
          *reinterpret_cast<f32x4*>(pp + 64 + c) = sb[q];
          if (EPI == 0) *reinterpret_cast<f32x4*>(pp + 128 + c) = kk[q];
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// 1x1 convolution with 64 input channels (K = 64: two MFMA K steps) and NC = 64 .. 512 output channels: the EXPANDING layers of the
// first residual stage (layer1.*.conv3 with the BatchNorm + ReLU of its input folded into the operand, layer1.0.downsample).  They are
// pure streaming -- 16 KiB in, NC/64 x 16 KiB out per 128 pixels, 0.1 us of MFMA -- and the 128-row kernel spends 10 us per 128 x 128
// tile on them (one tile per workgroup: prologue, two K tiles, epilogue; 3.0 TB/s).  Same skeleton as the 3x3 kernel above: weights
// resident in LDS in fragment order, a wave owns 128 pixels x ALL output channels, pixel fragments loaded straight into registers
// once per chunk (and transformed there: ATR 1 = bf16(relu(x*scale + shift2)), the arithmetic of conv_bf16.hip's operand transform),
// then NC/64 passes of 2 x 32 MFMAs over the same fragments, each with the c64 epilogue (permlane16_swap -> 16-byte stores, BatchNorm
// chunk partials of its 64 channels from the fp32 accumulators).  No barrier, no LDS traffic for the pixels; the HBM queue is kept
// full by the 8 independent waves per CU.
// KS = K / 32: 2 (64 input channels: 8 waves, two per SIMD) or 4 (128 input channels, `layer2.*.conv3`: the chunk's fragments are 128
// registers, so 4 waves per workgroup, one per SIMD with the whole register file; the weights of 512 x 128 fill 128 KiB of LDS).
// ATR 2 (KS = 2, NC = 64): the STEM of the bf16 trunks' 1-channel (OCT) encoder.  The 7x7 / stride 2 / pad 3 convolution is a
// 4x4 / stride 1 / pad 2 convolution over the 2x2 space-to-depth image xs [N][Hs][Ws][4] fp32 (edrl_space_to_depth2_f32,
// edrl_stem_weight_fold_f32): K = 4 window rows x 16 contiguous floats = 64.  `src` is that fp32 image; a lane's fragment of K step
// ks is half a window row -- 8 consecutive floats (two image columns x 4 channels), rounded to bf16 in registers; window rows /
// columns outside the image are zeros.  Same streaming skeleton as the 1x1 layers: 1.6 KiB of image in, 16 KiB out per 128 pixels.
struct StemGeom {
  int Hs, Ws;
  unsigned mg_w, mg_h;     // x / d == (x * mg) >> sh for x < 2^31 (conv_geom.h gather_magic)
  int sh_w, sh_h;
};

template <int ATR, int KS>
__global__ __launch_bounds__(KS == 2 ? 512 : 256, KS == 2 ? 2 : 1) void conv1x1_k64_bf16_kernel(
    const __bf16* __restrict__ src, const __bf16* __restrict__ w, __bf16* __restrict__ dst, int M, int NC, int nchunks,
    const float* __restrict__ fcoef, float* __restrict__ stat_part, StemGeom sg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int KC = 32 * KS;                    // input channels
  constexpr int NW = KS == 2 ? 8 : 4;            // waves per workgroup
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nct = NC >> 4;
  // weights [NC][KC] -> LDS fragment order [ks][channel tile][lane]: lane l = (channel 16 ct + (l&15), k 32 ks + 8 (l>>4) .. +7)
  for (int idx = tid; idx < KS * nct * 64; idx += NW * 64) {
    const int l = idx & 63, ct = (idx >> 6) % nct, ks = (idx >> 6) / nct;
    *reinterpret_cast<c64_bf16x8*>(smem + idx * 16) =
        *reinterpret_cast<const c64_bf16x8*>(w + (long)(16 * ct + (l & 15)) * KC + 32 * ks + 8 * (l >> 4));
  }
  float* tco = reinterpret_cast<float*>(smem + NC * KC * 2);    // ATR 1: [scale KC][shift2 KC] of the input BatchNorm, behind the weights
  if constexpr (ATR == 1) {
    if (tid < 2 * KC) tco[tid] = fcoef[(tid < KC ? 2 : 4) * KC + (tid % KC)];
  }
  __syncthreads();
  const int pl = lane & 15, gq = lane >> 4;
  const bool even = (gq & 1) == 0;
  const int cb0 = even ? 4 * gq : 16 + 4 * (gq - 1);
  auto rowsum = [&](float x) {
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x118, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x114, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x112, 0xf, 0xf, true));
    x += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x111, 0xf, 0xf, true));
    return x;
  };
  const unsigned char* wl = smem + lane * 16;
  for (int chunk = blockIdx.x * NW + wave; chunk < nchunks; chunk += gridDim.x * NW) {
    const int m0 = chunk * 128;
    // ---- the chunk's pixel fragments: 8 column tiles x KS K steps, 16 bytes per lane each
    c64_bf16x8 a[8][KS];
    if constexpr (ATR == 2) {
      // Branch-free gather: every load goes through a buffer descriptor based at the chunk's first image; a window row / column
      // outside the image (or a pixel past the end) is an out-of-range offset, which the hardware range check reads as zeros -- all
      // 32 loads of the chunk are in flight together (the first version predicated each load: one exposed latency per load).
      const float* xs = reinterpret_cast<const float*>(src);
      const int mu = __builtin_amdgcn_readfirstlane(m0);
      const unsigned r0 = (unsigned)(((unsigned long long)(unsigned)mu * sg.mg_w) >> sg.sh_w);
      const unsigned n0img = (unsigned)(((unsigned long long)r0 * sg.mg_h) >> sg.sh_h);
      const long imgb = (long)sg.Hs * sg.Ws * 16;                     // bytes per image
      long remb = (long)M * 16 - (long)n0img * imgb;
      if (remb > 0x7fffffffL) remb = 0x7fffffffL;
      const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)(xs + (long)n0img * sg.Hs * sg.Ws * 4), 0, (int)remb, 0x00020000);
      constexpr unsigned OOBX = 0x80000000u;
      f32x4 lo[8][KS], hi[8][KS];
#pragma unroll
      for (int pt = 0; pt < 8; ++pt) {
        const int m = m0 + 16 * pt + pl;
        const unsigned r = (unsigned)(((unsigned long long)(unsigned)m * sg.mg_w) >> sg.sh_w);      // m / Ws = n * Hs + oh
        const int ow = m - (int)r * sg.Ws;
        const unsigned nimg = (unsigned)(((unsigned long long)r * sg.mg_h) >> sg.sh_h);
        const int oh = (int)r - (int)nimg * sg.Hs;
        const int rowb = (int)(nimg - n0img) * sg.Hs;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const int qp = 4 * ks + gq;                       // k chunk of 8: window row qp >> 1, columns 2 (qp & 1), 2 (qp & 1) + 1
          const int ih = oh - 2 + (qp >> 1), iw = ow - 2 + 2 * (qp & 1);
          const bool rowok = m < M && (unsigned)ih < (unsigned)sg.Hs;
          const unsigned off = (unsigned)(((rowb + ih) * sg.Ws + iw) * 16);
          const unsigned o0 = (rowok && (unsigned)iw < (unsigned)sg.Ws) ? off : OOBX;
          const unsigned o1 = (rowok && (unsigned)(iw + 1) < (unsigned)sg.Ws) ? off + 16u : OOBX;
          lo[pt][ks] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)o0, 0, 0));
          hi[pt][ks] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)o1, 0, 0));
        }
      }
#pragma unroll
      for (int pt = 0; pt < 8; ++pt)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          c64_bf16x8 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = (__bf16)lo[pt][ks][e]; o[4 + e] = (__bf16)hi[pt][ks][e]; }
          a[pt][ks] = o;
        }
    } else {
#pragma unroll
    for (int pt = 0; pt < 8; ++pt) {
      const int m = m0 + 16 * pt + pl;
      const long mo = (long)(m < M ? m : M - 1) * KC;            // (rows past the end: a valid address, results never stored)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) a[pt][ks] = *reinterpret_cast<const c64_bf16x8*>(src + mo + 32 * ks + 8 * gq);
    }
    }
    if constexpr (ATR == 1) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        f32x4 tsc[2], tsh[2];            // scale / shift2 of this lane's input channels 32 ks + 8 gq + 4 h .. +3 (from LDS: no registers held)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          tsc[h] = *reinterpret_cast<const f32x4*>(tco + 32 * ks + 8 * gq + 4 * h);
          tsh[h] = *reinterpret_cast<const f32x4*>(tco + KC + 32 * ks + 8 * gq + 4 * h);
        }
#pragma unroll
        for (int pt = 0; pt < 8; ++pt) {
          c64_bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            o[e] = (__bf16)fmaxf(__builtin_fmaf((float)a[pt][ks][e], tsc[e >> 2][e & 3], tsh[e >> 2][e & 3]), 0.f);
          a[pt][ks] = o;
        }
      }
      __builtin_amdgcn_sched_barrier(0);       // (keeps the transform's temporaries out of the pass loop's register budget)
    }
#pragma unroll 1
    for (int q = 0; q < (NC >> 6); ++q) {
      f32x4 sa[4], sb[4], kk[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) sa[j] = sb[j] = kk[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        f32x4 acc[4][4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int pt = 0; pt < 4; ++pt) acc[ct][pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          c64_bf16x8 wf[4];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) wf[ct] = *reinterpret_cast<const c64_bf16x8*>(wl + ((ks * nct + 4 * q + ct) << 10));
#pragma unroll
          for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
              acc[ct][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ct], a[4 * blk + pt][ks], acc[ct][pt], 0, 0, 0);
        }
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
          f32x4 v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const auto p0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[0][pt][e]), __float_as_uint(acc[1][pt][e]), false, false);
            const auto p1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(acc[2][pt][e]), __float_as_uint(acc[3][pt][e]), false, false);
            v[0][e] = __uint_as_float(p0[0]); v[1][e] = __uint_as_float(p0[1]);
            v[2][e] = __uint_as_float(p1[0]); v[3][e] = __uint_as_float(p1[1]);
          }
          const int m = m0 + 64 * blk + 16 * pt + pl;
          const bool ok = m < M;
          if (stat_part) {
            if (blk == 0 && pt == 0) {
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) kk[j][e] = __shfl(v[j][e], lane & 48, 64);      // the chunk's first row
            }
            if (ok) {
#pragma unroll
              for (int j = 0; j < 4; ++j) { const f32x4 d = v[j] - kk[j]; sa[j] += d; sb[j] = __builtin_elementwise_fma(d, d, sb[j]); }
            }
          }
          if (ok) {
            c64_bf16x8 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o0[e] = (__bf16)v[0][e]; o0[4 + e] = (__bf16)v[1][e];
              o1[e] = (__bf16)v[2][e]; o1[4 + e] = (__bf16)v[3][e];
            }
            __bf16* dp = dst + (long)m * NC + 64 * q + cb0;
            *reinterpret_cast<c64_bf16x8*>(dp) = o0;
            *reinterpret_cast<c64_bf16x8*>(dp + 32) = o1;
          }
        }
      }
      if (stat_part) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) { sa[j][e] = rowsum(sa[j][e]); sb[j][e] = rowsum(sb[j][e]); }
        if (pl == 15) {
          float* pp = stat_part + (long)chunk * 3 * NC + 64 * q;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int c = cb0 + 4 * (j & 1) + 32 * (j >> 1);
            *reinterpret_cast<f32x4*>(pp + c) = sa[j];
            *reinterpret_cast<f32x4*>(pp + NC + c) = sb[j];
            *reinterpret_cast<f32x4*>(pp + 2 * (long)NC + c) = kk[j];
          }
        }
      }
    }
  }
}

bool conv1x1_k64_ok(int N, int H, int W, int Ci, int Co, int KH, int KW, int stride, int pad) {
  const int mode = edrl_cfg().bf16_k64;                // EDRL_BF16_K64: 0 off, 1 auto (default), 2 wherever the geometry allows
  if (mode == 0) return false;
  if ((Ci != 64 && Ci != 128) || (Co % 64) || Co < 64 || Co > 512 || KH != 1 || KW != 1 || stride != 1 || pad != 0 || N <= 0) return false;
  const long M = (long)N * H * W;
  if (M > 0x7fffff00L) return false;
  return mode == 2 || (M >= 128L * 2048 && Co >= 128);     // streaming layers only: below that the 128-row kernel's grid fills the chip
}

template <int ATR, int KS>
static void launch_k64_impl(const void* x, const float* in_fcoef, const void* w, void* y, long M, int Co, int nchunks, float* stat_part,
                            hipStream_t st, StemGeom sg = StemGeom{0, 0, 0u, 0u, 0, 0}) {
  constexpr int NW = KS == 2 ? 8 : 4;
  int grid = (nchunks + NW - 1) / NW;
  if (grid > 256) grid = 256;
  const int lds = Co * 32 * KS * 2 + 1024;
  auto kern = conv1x1_k64_bf16_kernel<ATR, KS>;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 512 * 32 * KS * 2 + 1024); attr = true; }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, st, (const __bf16*)x, (const __bf16*)w, (__bf16*)y, (int)M, Co, nchunks,
                     in_fcoef, stat_part, sg);
}

// y [M][Co] = conv1x1(x [M][Ci] (ATR: relu(x*scale + shift2) with in_fcoef [5][Ci]), w [Co][Ci]), Ci = 64 | 128; stat_part optional.
int launch_conv1x1_k64(const void* x, const float* in_fcoef, const void* w, void* y, int N, int H, int W, int Ci, int Co,
                       float* stat_part, hipStream_t st) {
  const long M = (long)N * H * W;
  const int nchunks = (int)((M + 127) / 128);
  if (((uintptr_t)x & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return EDRL_EINVAL;
  if (Ci == 64) {
    if (in_fcoef) launch_k64_impl<1, 2>(x, in_fcoef, w, y, M, Co, nchunks, stat_part, st);
    else launch_k64_impl<0, 2>(x, in_fcoef, w, y, M, Co, nchunks, stat_part, st);
  } else {
    if (in_fcoef) launch_k64_impl<1, 4>(x, in_fcoef, w, y, M, Co, nchunks, stat_part, st);
    else launch_k64_impl<0, 4>(x, in_fcoef, w, y, M, Co, nchunks, stat_part, st);
  }
  EDRL_LAUNCH_CHECK();
  return 0;
}

// Stem of a 1-channel bf16 trunk (kernel comment, ATR 2): xs fp32 [N][Hs][Ws][4] (space-to-depth image), w bf16 [64][64] (folded
// 4x4x4 weights, k = (window row, column, channel)), y bf16 [N][Hs][Ws][64], stat_part [ceil(M/128)][3][64] optional.
bool stem_s2d_bf16_ok(int N, int Hs, int Ws) {
  return N > 0 && Hs >= 2 && Ws >= 2 && (long)N * Hs * Ws < 0x7fffff00L && (long)N * Hs * Ws * 16 < (1L << 40);
}
int launch_stem_s2d_bf16(const float* xs, const void* w, void* y, int N, int Hs, int Ws, float* stat_part, hipStream_t st) {
  if (!stem_s2d_bf16_ok(N, Hs, Ws) || ((uintptr_t)xs & 15) || ((uintptr_t)w & 15) || ((uintptr_t)y & 15)) return EDRL_EINVAL;
  const long M = (long)N * Hs * Ws;
  StemGeom sg;
  sg.Hs = Hs; sg.Ws = Ws;
  gather_magic((unsigned)Ws, &sg.mg_w, &sg.sh_w);
  gather_magic((unsigned)Hs, &sg.mg_h, &sg.sh_h);
  launch_k64_impl<2, 2>(xs, nullptr, w, y, M, 64, (int)((M + 127) / 128), stat_part, st, sg);
  EDRL_LAUNCH_CHECK();
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 1-channel stem on the bf16 matrix pipe: dW [64 co][64 k] = sum_pixels d_raw[px][co] * patch[px][k], with
// patch[px][k] the 4x4 window of the space-to-depth fp32 image (k = (window row, column, channel), the forward's K order) rounded to
// bf16 in registers, d_raw bf16 [M][64].  3.3 GB of gradient + 0.4 GB of image per 2048 224x224 slices against 0.2 TFLOP: streaming.
// 256 threads, 32 KiB of LDS: both 128-pixel tiles go to LDS as [pixel][64] images (16-byte chunks XOR-swizzled so that the
// transposing fragment reads are conflict-free), each of the 4 waves owns one 32 x 32 block of dW (8 v_mfma_f32_32x32x16_bf16 per
// tile), accumulators live in registers across the workgroup's tiles, one [64][64] fp32 slab per workgroup (ordered reduction).
typedef short sw_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int sw_swz(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__device__ __forceinline__ c64_bf16x8 sw_tr_frag(const unsigned char* img, int pix0, int col0, int lane) {
  const int g16 = lane & 15, q = g16 >> 2, p4 = g16 & 3;
  const int col = col0 + 4 * p4;
  const int chunk = col >> 3, inner = (col & 7) * 2;
  const int r0 = pix0 + q, r1 = r0 + 4;
  const sw_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) sw_s16x4*)(img + r0 * 128 + ((chunk ^ sw_swz(r0)) * 16) + inner));
  const sw_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) sw_s16x4*)(img + r1 * 128 + ((chunk ^ sw_swz(r1)) * 16) + inner));
  union { struct { sw_s16x4 l, h; } s; c64_bf16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
}

__global__ __launch_bounds__(256, 2) void stem_wgrad_s2d_bf16_kernel(const __bf16* __restrict__ dy, const float* __restrict__ xs,
                                                                     float* __restrict__ slabs, int M, int ntiles, StemGeom sg) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 128 * 128];
  unsigned char* dimg = smem;                 // d_raw tile [128 px][64 co] bf16
  unsigned char* pimg = smem + 128 * 128;     // patch tile [128 px][64 k] bf16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cob = wave >> 1, kb = wave & 1;   // this wave's 32 x 32 block of dW
  const int lh = lane >> 5, li = lane & 31, cg = 16 * ((lane >> 4) & 1);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  constexpr unsigned OOBX = 0x80000000u;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int m0 = tile * 128;
    long rows = (long)M - m0; if (rows > 128) rows = 128;
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (long)m0 * 64), 0, (int)(rows * 128), 0x00020000);
    const unsigned r0u = (unsigned)(((unsigned long long)(unsigned)m0 * sg.mg_w) >> sg.sh_w);
    const unsigned n0img = (unsigned)(((unsigned long long)r0u * sg.mg_h) >> sg.sh_h);
    const long imgb = (long)sg.Hs * sg.Ws * 16;
    long remb = (long)M * 16 - (long)n0img * imgb;
    if (remb > 0x7fffffffL) remb = 0x7fffffffL;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(xs + (long)n0img * sg.Hs * sg.Ws * 4), 0, (int)remb, 0x00020000);
    // ---- the tile's pieces: 4 of d_raw (16 bytes) and 4 of the patch matrix (8 floats = half a window row) per thread
    c64_bf16x8 dv[4];
    f32x4 lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + 256 * i, px = id >> 3, ch = id & 7;
      dv[i] = __builtin_bit_cast(c64_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rd, (px * 64 + ch * 8) * 2, 0, 0));   // rows >= M: zeros
      const int m = m0 + px;
      const unsigned r = (unsigned)(((unsigned long long)(unsigned)m * sg.mg_w) >> sg.sh_w);
      const int ow = m - (int)r * sg.Ws;
      const unsigned nimg = (unsigned)(((unsigned long long)r * sg.mg_h) >> sg.sh_h);
      const int oh = (int)r - (int)nimg * sg.Hs;
      const int ih = oh - 2 + (ch >> 1), iw = ow - 2 + 2 * (ch & 1);
      const bool rowok = m < M && (unsigned)ih < (unsigned)sg.Hs;
      const unsigned off = (unsigned)((((int)(nimg - n0img) * sg.Hs + ih) * sg.Ws + iw) * 16);
      const unsigned o0 = (rowok && (unsigned)iw < (unsigned)sg.Ws) ? off : OOBX;
      const unsigned o1 = (rowok && (unsigned)(iw + 1) < (unsigned)sg.Ws) ? off + 16u : OOBX;
      lo[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)o0, 0, 0));
      hi[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)o1, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int id = tid + 256 * i, px = id >> 3, ch = id & 7;
      const int slot = (ch ^ sw_swz(px)) * 16;
      *reinterpret_cast<c64_bf16x8*>(dimg + px * 128 + slot) = dv[i];
      c64_bf16x8 pv;
#pragma unroll
      for (int e = 0; e < 4; ++e) { pv[e] = (__bf16)lo[i][e]; pv[4 + e] = (__bf16)hi[i][e]; }
      *reinterpret_cast<c64_bf16x8*>(pimg + px * 128 + slot) = pv;
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const int pix0 = 16 * ks + 8 * lh;
      const c64_bf16x8 af = sw_tr_frag(dimg, pix0, 32 * cob + cg, lane);
      const c64_bf16x8 bf = sw_tr_frag(pimg, pix0, 32 * kb + cg, lane);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  float* slab = slabs + (long)blockIdx.x * 64 * 64;
#pragma unroll
  for (int r = 0; r < 16; ++r) slab[(32 * cob + (r & 3) + 8 * (r >> 2) + 4 * lh) * 64 + 32 * kb + li] = acc[r];
}

static int stem_wgrad_grid(long M) {
  const long nt = (M + 127) / 128;
  return (int)(nt < 1024 ? nt : 1024);
}
int stem_wgrad_s2d_bf16_splits(int N, int Hs, int Ws) { return stem_wgrad_grid((long)N * Hs * Ws); }
size_t stem_wgrad_s2d_bf16_workspace_bytes(int N, int Hs, int Ws) { return (size_t)stem_wgrad_grid((long)N * Hs * Ws) * 64 * 64 * sizeof(float); }
// dy bf16 [N,Hs,Ws,64], xs fp32 [N,Hs,Ws,4] -> slabs [splits][64][64] fp32 (the caller reduces them in order)
int launch_stem_wgrad_s2d_bf16(const void* dy, const float* xs, float* slabs, int N, int Hs, int Ws, hipStream_t st) {
  if (!stem_s2d_bf16_ok(N, Hs, Ws) || ((uintptr_t)dy & 15) || ((uintptr_t)xs & 15) || ((uintptr_t)slabs & 15)) return EDRL_EINVAL;
  const long M = (long)N * Hs * Ws;
  StemGeom sg;
  sg.Hs = Hs; sg.Ws = Ws;
  gather_magic((unsigned)Ws, &sg.mg_w, &sg.sh_w);
  gather_magic((unsigned)Hs, &sg.mg_h, &sg.sh_h);
  hipLaunchKernelGGL(stem_wgrad_s2d_bf16_kernel, dim3(stem_wgrad_grid(M)), dim3(256), 0, st, (const __bf16*)dy, xs, slabs, (int)M,
                     (int)((M + 127) / 128), sg);
  EDRL_LAUNCH_CHECK();
  return 0;
}

static int c64_ceil_log2(unsigned d) {
  int s = 0;
  while ((1u << s) < d) ++s;
  return s;
}

bool conv3x3_c64_ok(int N, int H, int W, int Ci, int Co, int KH, int KW, int stride, int pad) {
  const int mode = edrl_cfg().bf16_c64;                // EDRL_BF16_C64: 0 off, 1 auto (default), 2 wherever the geometry allows
  if (mode == 0) return false;
  if (Ci != 64 || Co != 64 || KH != 3 || KW != 3 || stride != 1 || pad != 1 || N <= 0 || H < 2 || W < 2) return false;
  const long M = (long)N * H * W;
  if (M * 128 >= (1L << 31)) return false;             // one buffer descriptor over the tensor, 24-bit pixel indices
  return mode == 2 || M >= 128L * 256;                 // enough 128-pixel chunks to occupy the chip
}

// src [N,H,W,64] bf16; w: forward [Co][3][3][Ci], data gradient (flip = 1) the permuted [Ci][3][3][Co]; dst [N,H,W,64] bf16.
// stat_part (forward, optional): [ceil(M/128)][3][64]; ep_x / ep_mask / ep_part (data gradient with epilogue): see the kernel.
int launch_conv3x3_c64(const void* src, const void* w, int flip, void* dst, int N, int H, int W, float* stat_part, const void* ep_x,
                       const unsigned char* ep_mask, float* ep_part, const float* ep_mean, hipStream_t st) {
  C64Geom g;
  g.M = (int)((long)N * H * W);
  g.H = H; g.W = W;
  g.nchunks = (g.M + 127) / 128;
  g.k_w = 24 + c64_ceil_log2((unsigned)W);            // exact for numerators < 2^24 (M * 128 < 2^31)
  g.m_w = (unsigned)(((1ull << g.k_w) + (unsigned)W - 1) / (unsigned)W);
  g.k_h = 24 + c64_ceil_log2((unsigned)H);
  g.m_h = (unsigned)(((1ull << g.k_h) + (unsigned)H - 1) / (unsigned)H);
  if (((uintptr_t)src & 15) || ((uintptr_t)w & 15) || ((uintptr_t)dst & 15) || ((uintptr_t)ep_x & 15) || ((uintptr_t)ep_mask & 1))
    return EDRL_EINVAL;
  int grid = (g.nchunks + 7) / 8;
  if (grid > 256) grid = 256;
  if (ep_x) {
    if (!ep_mask || !ep_part || !ep_mean) return EDRL_EINVAL;
    auto kern = conv3x3_c64_bf16_kernel<1>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS); attr = true; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C64_LDS, st, (const __bf16*)src, (const __bf16*)w, (__bf16*)dst, g, flip, stat_part,
                       (const __bf16*)ep_x, ep_mask, ep_part, ep_mean);
  } else {
#ifdef EDRL_DIAG
    const int dbg = edrl_cfg().diag_c64;      // diagnostic builds (wrong outputs; libedrl_hip_diag.so only): 1 no tap shifts, 2 no loads
    if (dbg == 1 || dbg == 2) {
      auto kd = dbg == 1 ? conv3x3_c64_bf16_kernel<0, 1> : conv3x3_c64_bf16_kernel<0, 2>;
      (void)hipFuncSetAttribute((const void*)kd, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS);
      hipLaunchKernelGGL(kd, dim3(grid), dim3(512), C64_LDS, st, (const __bf16*)src, (const __bf16*)w, (__bf16*)dst, g, flip, stat_part,
                         (const __bf16*)nullptr, (const unsigned char*)nullptr, (float*)nullptr, (const float*)nullptr);
      EDRL_LAUNCH_CHECK();
      return 0;
    }
#endif
    auto kern = conv3x3_c64_bf16_kernel<0>;
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS); attr = true; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), C64_LDS, st, (const __bf16*)src, (const __bf16*)w, (__bf16*)dst, g, flip, stat_part,
                       (const __bf16*)nullptr, (const unsigned char*)nullptr, (float*)nullptr, (const float*)nullptr);
  }
  EDRL_LAUNCH_CHECK();
  return 0;
}
