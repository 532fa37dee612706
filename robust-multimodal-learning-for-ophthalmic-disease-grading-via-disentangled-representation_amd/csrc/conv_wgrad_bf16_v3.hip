// bf16 convolution weight gradient, generation 3 (the ResNet trunks' backward): the plain
// (unfused) weight gradients of the wide layers -- stages 3-4 of ResNet-50, >= 256 channels on both sides -- on the structure of
// conv_bf16_v3.hip: 256 x 256 output tile, 8 waves, v_mfma_f32_16x16x32_bf16, both operands moved global -> LDS by LDS-DMA in a
// 4-slot ring of 32-pixel units whose loads stay in flight across the workgroup barriers.
//
//     dW[co][kcol] (fp32) = sum_pixels dY[pix][co] * Xcol[pix][kcol],   kcol = (tap, ci)
//
// The contraction runs over PIXELS, the slow axis of both NHWC operands, and the MFMA wants 8 consecutive k per lane, so the
// fragments are fetched with the transposing LDS read ds_read_b64_tr_b16 (a 4-pixel x 16-channel block, delivered column-major).
// The DMA writes LDS lane-linearly (1 KiB per wave instruction), so the image is shaped on the SOURCE side: one piece = 16 pixel
// rows x 64 bytes (32 channels), lane l -> pixel row l>>2, 16-byte slot l&3 holding channel chunk (l&3) ^ 2*((l>>5)&1).  A 32-lane
// half of a transposed read takes pixel rows {0-3, 8-11} (or {4-7, 12-15}) of one piece at two chunks: with the XOR the 32 8-byte
// accesses fall on 32 distinct bank pairs (conflict-free; the same for both operands).  An operand unit is 16 pieces: 8 channel
// pairs (2 MFMA tiles each) x 2 pixel halves; wave w moves channel pair w of both operands (4 pieces per unit, as the forward core).
//
// dY rows are linear in the pixel index (per-lane offset + a uniform step per unit).  X rows are gathered (tap shift, padding,
// stride, image wrap): the row -> byte-offset decode of a unit's 32 pixels is done ONCE per workgroup by one wave (rotating, with
// host-made magic divisors; out-of-image taps and rows past the split become out-of-range offsets = the DMA writes zeros) into a
// 4-entry table ring in LDS, three units ahead; every wave adds its channel offset to the table entry of its rows.  A 256-column
// tile lies inside one tap (Ci % 256 == 0, host-checked), so the tap is uniform per workgroup.
//
// Split-K over pixel ranges: (Co/256)*(Ktot/256) tiles x `splits` <= 256 workgroups (one per CU, one round), fp32 partial slabs
// [split][Co][Ktot] reduced in fixed order by splitk_reduce_h_kernel (deterministic), as the 128x128 kernel of conv_bf16.hip.
#include "edrl_common.h"
#include "edrl_config.h"
#include <stdlib.h>
#include <string.h>
#include <type_traits>
#include "conv_bf16_v3.h"
#include "lds_dma.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short w3_s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* w3_lds_ptr_t;

#define W3_BK 32
#define W3_OPBYTES 16384                  // one operand unit: 32 pixels x 256 channels
#define W3_UNIT (2 * W3_OPBYTES)
#define W3_SLOTS 4
#define W3_TBL (W3_SLOTS * W3_UNIT)       // table ring: [4][32] byte offsets
#define W3_LDS (W3_TBL + 4 * 32 * 4)

struct WgradV3Geom {
  long P;
  int OH, OW, Co, SH, SW, SC, KH, KW, stride, pad, Ktot;
  int units_per_split, tiles_m, tiles_n;
  unsigned m_ohw, m_ow;       // magic multipliers: x / d == (x * m) >> k for x < 2^24
  int k_ohw, k_ow;
};

__device__ __forceinline__ bf16x8 w3_tr_frag(const unsigned char* s) {
  const w3_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3_s16x4*)(s));
  const w3_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) w3_s16x4*)(s + 256));
  union { struct { w3_s16x4 l, h; } p; bf16x8 v; } u;
  u.p.l = lo; u.p.h = hi;
  return u.v;
}

template <int DBG = 0>
__global__ __launch_bounds__(512, 2) void conv_wgrad_bf16_v3_kernel(const __bf16* __restrict__ dy, const __bf16* __restrict__ x,
                                                                    float* __restrict__ part, WgradV3Geom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int TM = 8, TN = 4;       // wave tile: 128 output channels (dY operand) x 64 kernel columns (X operand)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;
  // the workgroups of one split stream the same pixel range: consecutive logical ids, one XCD (one L2)
  const int lid = edrl_xcd_remap(blockIdx.x, gridDim.x);
  const int per_split = g.tiles_m * g.tiles_n;
  const int split = lid / per_split;
  const int trem = lid - split * per_split;
  const int tyi = trem / g.tiles_n;
  const int co0 = tyi * 256, n0 = (trem - tyi * g.tiles_n) * 256;
  const long units_total = (g.P + W3_BK - 1) / W3_BK;
  const long u_begin = (long)split * g.units_per_split;
  long u_end = u_begin + g.units_per_split;
  if (u_end > units_total) u_end = units_total;
  const int KU = (int)(u_end - u_begin);
  const long p_lo = u_begin * W3_BK;
  long p_hi = u_end * W3_BK; if (p_hi > g.P) p_hi = g.P;
  const int rows = (int)(p_hi - p_lo);
  const int ohw = g.OH * g.OW;
  const long n_lo = p_lo / ohw, n_hi = (p_hi - 1) / ohw;
  const unsigned ldy2 = (unsigned)g.Co * 2u, ldx2 = (unsigned)g.SC * 2u;
  const v3_i32x4 rs_y = v3_make_srd(dy + p_lo * g.Co, (unsigned)rows * ldy2);
  const v3_i32x4 rs_x = v3_make_srd(x + n_lo * g.SH * g.SW * g.SC, (unsigned)((n_hi - n_lo + 1) * g.SH * g.SW) * ldx2);
  const int tap = n0 / g.SC, kc0 = n0 - tap * g.SC;
  const int kh = tap / g.KW, kw = tap - kh * g.KW;
  const int prel0 = (int)(p_lo - n_lo * ohw);
  constexpr unsigned OOB = 0x80000000u;

  // ---- DMA addressing: wave w moves channel pair w (32 channels) of both operands, piece j = pixel half j
  const int drow = lane >> 2;
  const int dchunk = (lane & 3) ^ (((lane >> 5) & 1) << 1);
  unsigned aoff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) aoff[j] = (unsigned)(16 * j + drow) * ldy2 + (unsigned)(co0 + 32 * wave + 8 * dchunk) * 2u;
  const unsigned astep = W3_BK * ldy2;
  const unsigned bchan = (unsigned)(kc0 + 32 * wave + 8 * dchunk) * 2u;
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long)(w3_lds_ptr_t)smem) + (unsigned)wave * 2048u;
  unsigned* tbl = reinterpret_cast<unsigned*>(smem + W3_TBL);

  // byte offsets of the X rows of unit v (relative to the split's first image) -> table ring entry v & 3; lanes 0..31, one wave
  auto decode = [&](int v) {
    if (lane < 32) {
      const int r = v * W3_BK + lane;
      const unsigned pr = (unsigned)(prel0 + r);
      const unsigned n = (unsigned)(((unsigned long long)pr * g.m_ohw) >> g.k_ohw);
      const unsigned rem = pr - n * (unsigned)ohw;
      const unsigned oh = (unsigned)(((unsigned long long)rem * g.m_ow) >> g.k_ow);
      const unsigned ow = rem - oh * (unsigned)g.OW;
      const int ih = (int)oh * g.stride + kh - g.pad, iw = (int)ow * g.stride + kw - g.pad;
      const bool ok = r < rows && (unsigned)ih < (unsigned)g.SH && (unsigned)iw < (unsigned)g.SW;
      const unsigned off = ((n * (unsigned)g.SH + (unsigned)ih) * (unsigned)g.SW + (unsigned)iw) * ldx2;
      tbl[(v & 3) * 32 + lane] = ok ? off : OOB;
    }
  };
  auto issueA1 = [&](int slot, int j) {
    v3_dma16(lds0 + (unsigned)slot * W3_UNIT + (unsigned)j * 1024u, aoff[j], rs_y, 0);
  };
  auto issueB1 = [&](int slot, int j, unsigned voff) {
    v3_dma16(lds0 + (unsigned)slot * W3_UNIT + W3_OPBYTES + (unsigned)j * 1024u, voff, rs_x, 0);
  };

  // ---- fragment addressing (bytes inside an operand unit): group g4 = pixel octet, lane 4q+p of a group -> pixel row q, columns 4p..
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
  int lrd[2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
    lrd[t] = (g4 >> 1) * 1024 + (8 * (g4 & 1) + q) * 64 + ((((2 * t + (p4 >> 1)) ^ (2 * (g4 & 1)))) << 4) + 8 * (p4 & 1);
  const int a_base = (wave >> 2) * 8192, b_base = W3_OPBYTES + (wave & 3) * 4096;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ac[4], an[4], bc[TN], bn[TN];

  auto rdA = [&](int slot, int mh, bf16x8 (&af)[4]) {
    const unsigned char* s = smem + slot * W3_UNIT + a_base + mh * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = w3_tr_frag(s + lrd[i & 1] + (i >> 1) * 2048);
  };
  auto rdB = [&](int slot, bf16x8 (&bf)[TN]) {
    const unsigned char* s = smem + slot * W3_UNIT + b_base;
#pragma unroll
    for (int i = 0; i < TN; ++i) bf[i] = w3_tr_frag(s + lrd[i & 1] + (i >> 1) * 2048);
  };
  // 8 MFMAs: channel tiles 2q, 2q+1 of half MH against the 4 kernel-column tiles.  The X fragment is the first operand: a lane
  // ends up with 4 consecutive kernel columns of one output channel (row = 4*(lane>>4)+reg = column, col = lane&15 = channel).
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

  // one unit: `bcur` X fragments are multiplied, `bnxt` receives the next unit's (pipeline and wait counts as conv_bf16_v3.hip)
  auto unit = [&](int u, bf16x8 (&bcur)[TN], bf16x8 (&bnxt)[TN]) {
    const int slot = u & 3, nslot = (u + 3) & 3;
    __builtin_amdgcn_sched_barrier(0);
    rdA(slot, 1, an);
    if (wave == ((u + 3) & 7)) decode(u + 3);
    issueA1(nslot, 0);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H0{}, H0{}, ac, bcur);
    __builtin_amdgcn_sched_barrier(0);
    issueA1(nslot, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H0{}, H1{}, ac, bcur);
    __builtin_amdgcn_sched_barrier(0);
    // this wave's pieces of unit u+1 have landed (unit u+2: 4 pieces and the dY pieces of unit u+3: 2 may fly); its LDS reads and
    // the decoder's table writes are done
    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    const unsigned t0 = tbl[nslot * 32 + drow], t1 = tbl[nslot * 32 + 16 + drow];
    rdB((u + 1) & 3, bnxt);
    rdA((u + 1) & 3, 0, ac);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H1{}, H0{}, an, bcur);
    __builtin_amdgcn_sched_barrier(0);
    issueB1(nslot, 0, t0 + bchan);
    __builtin_amdgcn_sched_barrier(0);
    mma8(H1{}, H1{}, an, bcur);
    __builtin_amdgcn_sched_barrier(0);
    issueB1(nslot, 1, t1 + bchan);
    aoff[0] += astep; aoff[1] += astep;
  };

  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  if (KU > 0) {
    if (wave < 3) decode(wave);
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const unsigned t0 = tbl[u * 32 + drow], t1 = tbl[u * 32 + 16 + drow];
      issueA1(u, 0); issueA1(u, 1);
      issueB1(u, 0, t0 + bchan); issueB1(u, 1, t1 + bchan);
      aoff[0] += astep; aoff[1] += astep;
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");        // unit 0 landed (this wave's 4 pieces), units 1 and 2 still in flight
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
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the pipeline's tail pieces (zeros into consumed slots)
  }
  __builtin_amdgcn_sched_barrier(0);

  // ---- fp32 partial slab of this split: 16-byte stores, 4 lanes = 64 contiguous bytes of one output-channel row
  float* out = part + (long)split * g.Co * g.Ktot;
  const int fr = lane & 15;
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const long co = co0 + wm0 + 16 * j + fr;
#pragma unroll
    for (int i = 0; i < TN; ++i)
      *reinterpret_cast<f32x4*>(out + co * g.Ktot + n0 + wn0 + 16 * i + 4 * g4) = acc[i][j];
  }
}

static int w3_ceil_log2(unsigned d) {
  int s = 0;
  while ((1u << s) < d) ++s;
  return s;
}

static void wgrad_v3_plan(long P, int Co, int Ktot, int* splits, int* units_per_split) {
  const long tiles = (long)(Co / 256) * (Ktot / 256);
  const long units = (P + W3_BK - 1) / W3_BK;
  long s = tiles >= 256 ? 1 : 256 / tiles;      // one workgroup per CU, one round
  long by_len = units / 8; if (by_len < 1) by_len = 1;
  if (s > by_len) s = by_len;
  long ups = (units + s - 1) / s;
  s = (units + ups - 1) / ups;
  *splits = (int)s;
  *units_per_split = (int)ups;
}

bool wgrad_bf16_v3_ok(int N, int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int KH, int KW, int stride, int pad) {
  const int mode = edrl_cfg().bf16_wgrad_v3;           // EDRL_BF16_WGRAD_V3: 0 off, 1 auto (default), 2 force wherever the geometry allows
  if (mode == 0) return false;
  if ((Co % 256) || (Ci % 256) || N <= 0 || stride <= 0) return false;
  const long P = (long)N * Ho * Wo, ohw = (long)Ho * Wo;
  int splits, ups;
  wgrad_v3_plan(P, Co, KH * KW * Ci, &splits, &ups);
  const long span = ((long)ups + 4) * W3_BK;
  const bool can = ohw < (1L << 16) && span + ohw < (1L << 24) && span * Co * 2 < (1L << 31) &&
                   (span / ohw + 2) * Hi * Wi * Ci * 2 < (1L << 31);
  if (!can) return false;
  if (mode == 2) return true;
  // Measured against the 128x128 kernel (scripts/wgrad_layer_bench.py): +45-60 % at 2112 images, +25-35 % at 256, a tie at 64 images
  // (12.5 k pixels on the 14x14 maps: both are launch / latency bound there).
  return P >= 8192;
}

size_t wgrad_bf16_v3_workspace_bytes(int N, int Ho, int Wo, int Co, int Ci, int KH, int KW) {
  if ((Co % 256) || (Ci % 256)) return 0;
  int splits, ups;
  wgrad_v3_plan((long)N * Ho * Wo, Co, KH * KW * Ci, &splits, &ups);
  return (size_t)splits * Co * KH * KW * Ci * sizeof(float);
}

int launch_wgrad_bf16_v3(const void* dy, const void* x, float* workspace, size_t workspace_bytes, int N, int Hi, int Wi, int Ci,
                         int Ho, int Wo, int Co, int KH, int KW, int stride, int pad, int* splits_out, hipStream_t st) {
  WgradV3Geom g;
  g.P = (long)N * Ho * Wo;
  g.OH = Ho; g.OW = Wo; g.Co = Co; g.SH = Hi; g.SW = Wi; g.SC = Ci;
  g.KH = KH; g.KW = KW; g.stride = stride; g.pad = pad; g.Ktot = KH * KW * Ci;
  int splits;
  wgrad_v3_plan(g.P, Co, g.Ktot, &splits, &g.units_per_split);
  if (workspace == nullptr || workspace_bytes < (size_t)splits * Co * g.Ktot * sizeof(float)) return EDRL_ENOSPC;
  g.tiles_m = Co / 256; g.tiles_n = g.Ktot / 256;
  const unsigned ohw = (unsigned)(Ho * Wo);
  g.k_ohw = 24 + w3_ceil_log2(ohw);
  g.m_ohw = (unsigned)(((1ull << g.k_ohw) + ohw - 1) / ohw);
  g.k_ow = 24 + w3_ceil_log2((unsigned)Wo);
  g.m_ow = (unsigned)(((1ull << g.k_ow) + (unsigned)Wo - 1) / (unsigned)Wo);
  const long nblk = (long)g.tiles_m * g.tiles_n * splits;
  if (nblk > 0x7fffffffL) return EDRL_EINVAL;
  auto kern = conv_wgrad_bf16_v3_kernel<0>;
  static bool attr_set = false;
  if (!attr_set) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, W3_LDS); attr_set = true; }
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(512), W3_LDS, st, (const __bf16*)dy, (const __bf16*)x, workspace, g);
  EDRL_LAUNCH_CHECK();
  *splits_out = splits;
  return 0;
}
