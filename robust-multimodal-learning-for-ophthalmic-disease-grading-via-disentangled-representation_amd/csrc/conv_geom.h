// Geometry of the implicit-GEMM "gather" contraction shared by the fp32 and bf16 conv kernels.
#pragma once

struct GatherGeom {
  int M;            // destination rows = N*OH*OW
  int OH, OW;       // destination spatial
  int NC;           // destination channels (GEMM N)
  int SH, SW, SC;   // source spatial / channels
  int KH, KW, stride, pad;
  int Ktot;         // KH*KW*SC
  long ld_src;      // source pixel stride (elements)
  long ld_dst;      // destination pixel stride (elements)
  long ld_aux;      // pixel stride of `mul` / `addend` operands (elements)
  int flags;        // bit0 relu, bit1 accumulate into dst
  // Sub-lattice ("parity class") view used by the strided data gradient: rows enumerate the
  // destination pixels (h0 + i*step, w0 + j*step), i < OHs, j < OWs, and K runs over the taps
  // (kh0 + a*kstep, kw0 + b*kstep), a < KHs, b < KWs — exactly the taps that hit that class, so no
  // MFMA work is spent on masked taps.  The plain case is step = kstep = 1, h0 = w0 = kh0 = kw0 = 0.
  int h0, w0, step, OHs, OWs;
  int kh0, kw0, kstep, KHs, KWs;
  int Kfull;        // KH*KW*SC: row stride of the weight matrix
  int sshift;       // log2(stride) for DGRAD
  // Fused BatchNorm statistics (GF_STATS, forward only): every workgroup emits the shifted moments of its 128 output
  // rows, sum (y-K) and sum (y-K)^2 with K = the workgroup's first output row, into stat_part[tile_m][3][NC] (plane 2 = K) — the chunk
  // partial layout of bn_pool.hip, so the BN statistics need no extra pass over the conv output.
  float* stat_part;
  const float* stat_shift;
  // Division of a row index (< 2^31) by OHs*OWs and by OWs as multiply-high + shift: q = (n * mg) >> sh with sh = 31 + ceil(log2 d),
  // mg = ceil(2^sh / d) (exact for n < 2^31).  Filled by the launcher (gather_geom_magic); replaces ~100-instruction software
  // divisions in the row decode of every workgroup.
  unsigned mg_ohw, mg_ow;
  int sh_ohw, sh_ow;
  // Depth taps (conv_gather_f32_v2_kernel<..., VOL = true> only: 3-D convolution over NDHWC volumes, csrc/vol_ops.hip; every other
  // instantiation never reads these and their launchers leave them unset).  The row index enumerates (sample, do, oh, ow); K runs
  // over (kh, kw, kd, ci) -- depth tap innermost of the taps, i.e. the weight layout [Co][KH][KW][KD*Ci] of the depth-unfolded form.
  int KD, SD, OD, dstride, dpad;   // depth taps, source / destination depth, depth stride and padding
  unsigned mg_od;                  // image index / OD (magic division, as above)
  int sh_od;
};

static inline void gather_magic(unsigned d, unsigned* mg, int* sh) {
  int l = 0;
  while ((1ull << l) < d) ++l;
  *sh = 31 + l;
  *mg = (unsigned)(((1ull << *sh) + d - 1) / d);
}
static inline void gather_geom_magic(GatherGeom* g) {
  const long ohw = (long)g->OHs * g->OWs;
  gather_magic(ohw > 0 && ohw < (1L << 31) ? (unsigned)ohw : 1u, &g->mg_ohw, &g->sh_ohw);
  gather_magic(g->OWs > 0 ? (unsigned)g->OWs : 1u, &g->mg_ow, &g->sh_ow);
}

// Fused-BatchNorm operands of the gather kernels (conv_gather_*_kernel<..., ATR, EPI>).
struct GatherFuse {
  const void* src2;            // ATR 2: raw conv output paired with the masked gradient in `src` (same geometry and ld)
  const float* acoef;          // ATR 1: fcoef [5][SC] = {mean, rstd, scale, shift, shift2} of the producing BatchNorm
                               // ATR 2: bcoef [4][SC] = {A, nK2, C2, mean} of this layer's BatchNorm backward
  const void* ep_x;            // EPI 1: raw conv output the destination tensor is the (post-ReLU) gradient of
  long ld_ep;                  //        its pixel stride (elements)
  const unsigned char* ep_mask;//        ReLU sign bytes [pixel][NC/4], or NULL: recompute the decision from ep_x when GF_EPI_RELU
  const float* ep_fcoef;       //        fcoef [5][NC] of that BatchNorm
  float* ep_part;              //        [chunks][2][NC] partial sums (sum g, sum g*x), one chunk per 128-row tile
  int ep_chunk0;               //        first chunk of this launch (parity classes of a strided data gradient)
};

// K-split of the tail tiles of a launch (conv_gemm.hip gather_ksplit_plan; fp32 gather kernel only)
struct GatherSplit {
  int mode;                    // 0: off; 1: main launch (body tiles + the split workgroups); 2: fix-up launch over the tail tiles
  int n_body;                  // workgroups (= tiles) of the body, a multiple of 256
  int ksplit, kt_per;          // parts per tail tile, K tiles per part
  float* slab;                 // [tail tile][part][accumulator element][thread] fp32
};

#define GF_RELU 1
#define GF_ACCUM 2
#define GF_VEC_EPI 4   // set by the host when the float4 epilogue is legal (alignment, NC % 4 == 0)
#define GF_STATS 8     // emit BatchNorm chunk partials from the vector epilogue
#define GF_EPI_RELU 16 // EPI 1 without sign bytes: the BatchNorm whose backward is reduced was followed by a ReLU
#define GF_LEAN_STRIDED 32  // set by the launcher: the per-workgroup destination footprint of a strided parity class fits a 2 GiB descriptor
#define GF_KTAIL 128         // (bf16 128-row kernel, A/B switch EDRL_BF16_KTAIL=1) issue the load round past the last K tile as before
#define GF_EPI_VW4 64       // (bf16 128-row kernel, A/B switch EDRL_BF16_EPI_VW4=1) keep the 4-channel epilogue lanes
