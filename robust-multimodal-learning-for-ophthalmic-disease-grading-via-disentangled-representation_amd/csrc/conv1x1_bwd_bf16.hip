// Backward of the EXPANDING 1x1 layers of the first residual stage of the bf16 trunks (`layer1.*.conv3`: 64 -> 256 channels at
// 56x56 for 224x224 inputs) as ONE streaming kernel: data gradient AND weight
// gradient from a single pass over the two 256-channel tensors.
//
// Inside a fused-BatchNorm block (encoders._TrunkFn) this layer's backward needs
//     d3  = A*g + nK2*y + C2                 d_raw of bn3: g = masked gradient [M][256], y = raw conv3 output [M][256]
//     a2  = relu(x2*scale + shift2)          the layer's input: x2 = raw conv2 output [M][64]
//     dW  = d3^T a2                          [256][64] weight gradient
//     g2  = mask(bn2) * (d3 W)               [M][64] masked gradient of bn2's output, + (sum g2, sum g2*(x2 - mean)) partials
// Rounds 2-3 ran two kernels (conv_wgrad_bf16_kernel<DYT 2, XT 1> and conv_gather_bf16_kernel<DGRAD, ATR 2, EPI 1>), each reading g
// and y -- the two LARGE tensors (3.3 GB each per 2048 slices) -- and each already HBM-bound (5.1-5.5 TB/s).  Both contractions
// are tiny next to the bytes (0.2 us of MFMA per 128 pixels against 7.6 us of HBM time per CU), so the only lever is to read the
// pair once: 15.6 GB -> 8.2 GB per call.
//
// Skeleton (weight-stationary, persistent): 512 workgroups x 256 threads, TWO per CU (72 KiB of LDS and the whole register file of
// one wave per SIMD each: while one multiplies, the other converts and stages -- the first version, one 512-thread workgroup per CU
// on 128-pixel tiles, ran its phases in lockstep at 3.7 TB/s); the permuted weights Wt [64][256] sit in LDS in MFMA-fragment order
// for the whole launch.  Per 64-pixel tile:
//   * every thread requests the tile's 18 16-byte pieces (8 of g, 8 of y, 2 of x2) at once; while they fly the other workgroup
//     of the CU multiplies its tile (two workgroups x 72 KiB in flight keep the HBM queue full);
//   * d3 is formed once per element (fp32 fma, one rounding -- the arithmetic of conv_bf16.hip's ATR 2 operand) and written to an
//     LDS image [64 px][256 co] whose 16-byte chunks are XOR-swizzled with S(row) = ((row&3)<<2)|((row>>2)&3): conflict-free BOTH
//     for the row reads of the data gradient (ds_read_b128, 16 rows at one chunk per lane group) and for the transposing reads of the
//     weight gradient (ds_read_b64_tr_b16, 4 rows x 4 chunks per 32-lane half); a2 = relu(bn(x2)) is formed once per element and goes to LDS, [64][64], swizzled likewise;
//   * data gradient: wave w owns pixels 32 (w>>1) .. +31 x channels 32 (w&1) .. +31, K = 256: 16 v_mfma_f32_32x32x16_bf16 with the
//     weight fragment as first operand -> a lane holds 4 x 4 consecutive channels of ONE pixel; epilogue in registers: ReLU decision
//     of bn2 recomputed from the raw x2 values of its pixel (re-read from global: L2 hits; same fma as the forward), masked gradient stored (8-byte pieces), partial sums kept
//     per lane for the whole launch and reduced once at the end -> ep_part [2 * workgroup + pixel block][2][64];
//   * weight gradient: wave w owns output channels 64 w .. +63 x all 64 input channels, K = 64 pixels: 16 MFMAs on transposed
//     fragments of the d3 and a2 images; fp32 accumulators live in registers
//     across all tiles of the workgroup -> one [256][64] slab per workgroup, summed in fixed order by splitk_reduce_h_kernel.
// Deterministic: static tile -> workgroup assignment, ordered reductions, no atomics.
#include "edrl_common.h"
#include <stdlib.h>
#include <string.h>
#include "conv_bf16_v3.h"

typedef __bf16 b1_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 b1_bf16x4 __attribute__((ext_vector_type(4)));
typedef short b1_s16x4 __attribute__((ext_vector_type(4)));

#define B1_CI 64
#define B1_CO 256
#define B1_P 64                                    // pixels per tile
#define B1_NT 256                                  // threads per workgroup (4 waves; two workgroups per CU)
#define B1_W_BYTES (16 * 2 * 64 * 16)              // 32 KiB: Wt fragments [k step][channel tile][lane]
#define B1_D_BYTES (B1_P * B1_CO * 2)              // 32 KiB: d3 image
#define B1_X_BYTES (B1_P * B1_CI * 2)              // 8 KiB: a2 = relu(bn(x2)) image
#define B1_T_BYTES ((3 * B1_CO + 3 * B1_CI) * 4)   // coefficient tables: A, nK2, C2 [256]; scale, shift2, mean [64]
#define B1_LDS (B1_W_BYTES + B1_D_BYTES + B1_X_BYTES + B1_T_BYTES)

__device__ __forceinline__ int b1_sd(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }          // d3 image: 4-bit chunk XOR
__device__ __forceinline__ int b1_sx(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }   // x2 image: 3-bit chunk XOR

// transposing fragment read (two ds_read_b64_tr_b16): pixel rows pix0 .. pix0+7, this lane's column col0 + (lane & 15) of a
// [pixel][channel] image with `pitch` bytes per row whose 16-byte chunks are swizzled by `swz(row)` (low `bits` bits of the chunk)
template <int PITCH, bool DIMG>
__device__ __forceinline__ b1_bf16x8 b1_tr_frag(const unsigned char* img, int pix0, int col0, int lane) {
  const int g16 = lane & 15, q = g16 >> 2, p4 = g16 & 3;
  const int col = col0 + 4 * p4;                    // 4 consecutive columns = 8 bytes
  const int chunk = col >> 3, inner = (col & 7) * 2;
  const int r0 = pix0 + q, r1 = r0 + 4;
  const int s0 = DIMG ? ((chunk & 16) | ((chunk & 15) ^ b1_sd(r0))) : (chunk ^ b1_sx(r0));
  const int s1 = DIMG ? ((chunk & 16) | ((chunk & 15) ^ b1_sd(r1))) : (chunk ^ b1_sx(r1));
  const b1_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) b1_s16x4*)(img + r0 * PITCH + s0 * 16 + inner));
  const b1_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) b1_s16x4*)(img + r1 * PITCH + s1 * 16 + inner));
  union { struct { b1_s16x4 l, h; } s; b1_bf16x8 v; } u;
  u.s.l = lo; u.s.h = hi;
  return u.v;
}

__global__ __launch_bounds__(B1_NT, 2) void conv1x1_k64_bwd_bf16_kernel(
    const __bf16* __restrict__ gin, const __bf16* __restrict__ yraw, const float* __restrict__ bcoef,
    const __bf16* __restrict__ x2, const float* __restrict__ x2coef, const __bf16* __restrict__ wt, __bf16* __restrict__ g2,
    float* __restrict__ ep_part, float* __restrict__ dw_slabs, int M, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wl = smem;
  unsigned char* dimg = smem + B1_W_BYTES;
  unsigned char* ximg = dimg + B1_D_BYTES;
  float* tab = reinterpret_cast<float*>(ximg + B1_X_BYTES);      // [A 256][nK2 256][C2 256][scale 64][shift2 64][mean 64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- weights -> LDS in fragment order: fragment (k step ks, channel tile ct), lane l = (row ci = 32 ct + (l & 31),
  //      k = output channels 16 ks + 8 (l >> 5) .. +7) of Wt [64][256]
  for (int idx = tid; idx < 16 * 2 * 64; idx += B1_NT) {
    const int l = idx & 63, ct = (idx >> 6) & 1, ks = idx >> 7;
    *reinterpret_cast<b1_bf16x8*>(wl + idx * 16) =
        *reinterpret_cast<const b1_bf16x8*>(wt + (long)(32 * ct + (l & 31)) * B1_CO + 16 * ks + 8 * (l >> 5));
  }
  for (int i = tid; i < 3 * B1_CO; i += B1_NT) tab[i] = bcoef[i];                                        // rows 0..2 of bcoef [4][256]
  if (tid < B1_CI) {
    tab[3 * B1_CO + tid] = x2coef[2 * B1_CI + tid];                                                    // scale
    tab[3 * B1_CO + B1_CI + tid] = x2coef[4 * B1_CI + tid];                                            // shift2
    tab[3 * B1_CO + 2 * B1_CI + tid] = x2coef[tid];                                                    // mean
  }
  __syncthreads();
  const float* tA = tab, *tK = tab + B1_CO, *tC = tab + 2 * B1_CO;
  const float* tsc = tab + 3 * B1_CO, *tsh = tsc + B1_CI, *tmu = tsh + B1_CI;

  // ---- staging coordinates: d3 pieces (row = (tid >> 5) + 8 i, chunk = tid & 31), x2 pieces (id = tid + 256 i: row id >> 3, chunk id & 7)
  const int drow = tid >> 5, dchunk = tid & 31;
  b1_bf16x8 G[8], Y[8], X[2];
  auto issue = [&](int tile, int i) {             // piece i of g / y (i < 8) of `tile`: rows past M read as zeros (range check)
    const long m0 = (long)tile * B1_P;
    long rows = (long)M - m0; if (rows > B1_P) rows = B1_P; if (rows < 0) rows = 0;
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(gin + m0 * B1_CO), 0, (int)(rows * B1_CO * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(yraw + m0 * B1_CO), 0, (int)(rows * B1_CO * 2), 0x00020000);
    const int off = ((drow + 8 * i) * B1_CO + dchunk * 8) * 2;
    G[i] = __builtin_bit_cast(b1_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0));
    Y[i] = __builtin_bit_cast(b1_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(ry, off, 0, 0));
  };
  auto issue_x = [&](int tile, int i) {
    const long m0 = (long)tile * B1_P;
    long rows = (long)M - m0; if (rows > B1_P) rows = B1_P; if (rows < 0) rows = 0;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x2 + m0 * B1_CI), 0, (int)(rows * B1_CI * 2), 0x00020000);
    const int id = tid + B1_NT * i;
    X[i] = __builtin_bit_cast(b1_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rx, ((id >> 3) * B1_CI + (id & 7) * 8) * 2, 0, 0));
  };

  // ---- accumulators that live for the whole launch
  f32x16 accw[2][2];                              // weight gradient: rows 32 (2 wave + b) .. +31, columns 32 ct .. +31
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[b][c][r] = 0.f;
  const int pb = wave >> 1, dct = wave & 1;       // data gradient: pixel block, channel tile of this wave
  const int lh = lane >> 5, li = lane & 31;
  f32x4 es0[4], es1[4];                           // (sum g2, sum g2*(x2 - mean)) of this lane's 4 x 4 channels
#pragma unroll
  for (int j = 0; j < 4; ++j) es0[j] = es1[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const long m0 = (long)tile * B1_P;
    // the tile's 18 pieces per thread: all in flight together; the OTHER workgroup of the CU multiplies meanwhile (no register
    // prefetch across tiles: with 64 accumulator registers of the weight gradient it would not fit the 256-register budget)
#pragma unroll
    for (int i = 0; i < 8; ++i) issue(tile, i);
#pragma unroll
    for (int i = 0; i < 2; ++i) issue_x(tile, i);
    // raw x2 of this lane's epilogue pixel (4 x 4 channels): the LDS image holds the ACTIVATED tensor, the BatchNorm-backward sums
    // need the raw one (L2 hits: the workgroup reads these lines for the staging pieces in the same breath)
    b1_bf16x4 xe4[4];
    {
      const long m0e = (long)tile * B1_P;
      long rows = (long)M - m0e; if (rows > B1_P) rows = B1_P;
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x2 + m0e * B1_CI), 0, (int)(rows * B1_CI * 2), 0x00020000);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const auto v2 = __builtin_amdgcn_raw_buffer_load_b64(rx, ((32 * pb + li) * B1_CI + 32 * dct + 8 * j + 4 * lh) * 2, 0, 0);
        xe4[j] = __builtin_bit_cast(b1_bf16x4, v2);
      }
    }
    // ---- consume the tile's pieces: d3 -> LDS, x2 raw -> LDS; each register is re-issued for the next tile as soon as it is read
    f32x4 cA[2], cK[2], cC[2];                    // A, nK2, C2 of this thread's 8 output channels (from the LDS table: no registers held
#pragma unroll                                   //  across the multiply phase)
    for (int h = 0; h < 2; ++h) {
      cA[h] = *reinterpret_cast<const f32x4*>(tA + dchunk * 8 + 4 * h);
      cK[h] = *reinterpret_cast<const f32x4*>(tK + dchunk * 8 + 4 * h);
      cC[h] = *reinterpret_cast<const f32x4*>(tC + dchunk * 8 + 4 * h);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = drow + 8 * i;
      const bool ok = m0 + row < M;
      b1_bf16x8 d;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = __builtin_fmaf(cK[e >> 2][e & 3], (float)Y[i][e], __builtin_fmaf(cA[e >> 2][e & 3], (float)G[i][e], cC[e >> 2][e & 3]));
        d[e] = ok ? (__bf16)v : (__bf16)0.f;
      }

      const int slot = (dchunk & 16) | ((dchunk & 15) ^ b1_sd(row));
      *reinterpret_cast<b1_bf16x8*>(dimg + row * (B1_CO * 2) + slot * 16) = d;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {                 // a2 = relu(x2*scale + shift2), formed ONCE here (the weight gradient's operand)
      const int id = tid + B1_NT * i, row = id >> 3, chunk = id & 7;
      const f32x4 s0 = *reinterpret_cast<const f32x4*>(tsc + chunk * 8), s1 = *reinterpret_cast<const f32x4*>(tsc + chunk * 8 + 4);
      const f32x4 h0 = *reinterpret_cast<const f32x4*>(tsh + chunk * 8), h1 = *reinterpret_cast<const f32x4*>(tsh + chunk * 8 + 4);
      b1_bf16x8 av;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        av[e] = (__bf16)fmaxf(__builtin_fmaf((float)X[i][e], s0[e], h0[e]), 0.f);
        av[4 + e] = (__bf16)fmaxf(__builtin_fmaf((float)X[i][4 + e], s1[e], h1[e]), 0.f);
      }
      *reinterpret_cast<b1_bf16x8*>(ximg + row * (B1_CI * 2) + ((chunk ^ b1_sx(row)) * 16)) = av;
    }
    __syncthreads();

    // ---- data gradient: D[ci 32][px 32] = Wt[ci][co] d3[px][co]^T over 16 k steps
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    {
      const int prow = 32 * pb + li;
      const unsigned char* drowp = dimg + prow * (B1_CO * 2);
      const int sdr = b1_sd(prow);
#pragma unroll 4
      for (int ks = 0; ks < 16; ++ks) {
        const b1_bf16x8 wf = *reinterpret_cast<const b1_bf16x8*>(wl + ((ks * 2 + dct) * 64 + lane) * 16);
        const int chunk = 2 * ks + lh;
        const b1_bf16x8 df = *reinterpret_cast<const b1_bf16x8*>(drowp + (((chunk & 16) | ((chunk & 15) ^ sdr)) * 16));
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, df, acc, 0, 0, 0);
      }
    }
    // ---- epilogue of the data gradient, in registers: lane = pixel 32 pb + li, channels 32 dct + 8 j + 4 lh .. +3 (j = r >> 2)
    {
      const int prow = 32 * pb + li;
      const long m = m0 + prow;
      if (m < M) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int cb = 32 * dct + 8 * j + 4 * lh;       // scale / shift2 / mean of these 4 channels: LDS table
          const f32x4 esc = *reinterpret_cast<const f32x4*>(tsc + cb), esh = *reinterpret_cast<const f32x4*>(tsh + cb);
          const f32x4 emu = *reinterpret_cast<const f32x4*>(tmu + cb);
          const b1_bf16x4 xq = xe4[j];
          b1_bf16x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float xe = (float)xq[e];
            const float ve = __builtin_fmaf(xe, esc[e], esh[e]) > 0.f ? acc[4 * j + e] : 0.f;
            es0[j][e] += ve;
            es1[j][e] = __builtin_fmaf(ve, xe - emu[e], es1[j][e]);
            o[e] = (__bf16)ve;
          }
          *reinterpret_cast<b1_bf16x4*>(g2 + m * B1_CI + 32 * dct + 8 * j + 4 * lh) = o;
        }
      }
    }
    // ---- weight gradient: D[co 32][ci 32] += d3^T a2 over the tile's 64 pixels (4 k steps of 16); this wave: co blocks 2 w, 2 w + 1
#pragma unroll 2
    for (int ks = 0; ks < 4; ++ks) {
      const int pix0 = 16 * ks + 8 * lh;
      const int cg = 16 * ((lane >> 4) & 1);
      b1_bf16x8 bf[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) bf[c] = b1_tr_frag<B1_CI * 2, false>(ximg, pix0, 32 * c + cg, lane);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const b1_bf16x8 af = b1_tr_frag<B1_CO * 2, true>(dimg, pix0, 32 * (2 * wave + b) + cg, lane);
#pragma unroll
        for (int c = 0; c < 2; ++c) accw[b][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf[c], accw[b][c], 0, 0, 0);
      }
    }
    __syncthreads();                                // every read of the tile images is done: the next tile may overwrite them
  }

  // ---- weight-gradient slab of this workgroup: rows (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32, column 32 c + li
  float* slab = dw_slabs + (long)blockIdx.x * B1_CO * B1_CI;
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        slab[(32 * (2 * wave + b) + (r & 3) + 8 * (r >> 2) + 4 * lh) * B1_CI + 32 * c + li] = accw[b][c][r];
  // ---- BatchNorm-backward partial sums: reduce over the 32 pixel lanes of each half, one chunk per (workgroup, pixel block)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) {
        es0[j][e] += __shfl_xor(es0[j][e], o, 64);
        es1[j][e] += __shfl_xor(es1[j][e], o, 64);
      }
    }
  if (li == 0) {
    float* pp = ep_part + ((long)blockIdx.x * 2 + pb) * 2 * B1_CI;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<f32x4*>(pp + 32 * dct + 8 * j + 4 * lh) = es0[j];
      *reinterpret_cast<f32x4*>(pp + B1_CI + 32 * dct + 8 * j + 4 * lh) = es1[j];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- host side
static int b1_grid(long M) {
  const long nt = (M + B1_P - 1) / B1_P;
  return (int)(nt < 512 ? nt : 512);
}
bool conv1x1_k64_bwd_ok(int N, int H, int W, int Ci, int Co) {
  const long M = (long)N * H * W;
  return Ci == B1_CI && Co == B1_CO && N > 0 && M > 0 && M < 0x7fffff00L;
}
long conv1x1_k64_bwd_chunks(int N, int H, int W) { return 2L * b1_grid((long)N * H * W); }
size_t conv1x1_k64_bwd_workspace_bytes(int N, int H, int W) { return (size_t)b1_grid((long)N * H * W) * B1_CO * B1_CI * sizeof(float); }
int conv1x1_k64_bwd_splits(int N, int H, int W) { return b1_grid((long)N * H * W); }

int launch_conv1x1_k64_bwd(const void* g, const void* yraw, const float* bcoef, const void* x2, const float* x2coef, const void* wt,
                           void* g2, float* ep_part, float* dw_slabs, int N, int H, int W, hipStream_t st) {
  const long M = (long)N * H * W;
  if (((uintptr_t)g & 15) || ((uintptr_t)yraw & 15) || ((uintptr_t)x2 & 15) || ((uintptr_t)wt & 15) || ((uintptr_t)g2 & 7) ||
      ((uintptr_t)ep_part & 15))
    return EDRL_EINVAL;
  const int ntiles = (int)((M + B1_P - 1) / B1_P);
  auto kern = conv1x1_k64_bwd_bf16_kernel;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, B1_LDS); attr = true; }
  hipLaunchKernelGGL(kern, dim3(b1_grid(M)), dim3(B1_NT), B1_LDS, st, (const __bf16*)g, (const __bf16*)yraw, bcoef, (const __bf16*)x2,
                     x2coef, (const __bf16*)wt, (__bf16*)g2, ep_part, dw_slabs, (int)M, ntiles);
  EDRL_LAUNCH_CHECK();
  return 0;
}
