// Run-time switches of the library, read from the environment ONCE (first use) into one struct -- no getenv on the launch path,
// which autograd worker threads run concurrently.  edrl_config_reload() (C-ABI, include/edrl_hip.h) re-reads the environment: tests
// and the A/B scripts call it between launches after changing a switch (not concurrently with launches).
// Every switch is an A/B lever documented in DESIGN.md section 5; the defaults are the production configuration.
#pragma once

struct EdrlConfig {
  // bf16 trunk (C2/C4)
  int bf16_v3;           // EDRL_BF16_V3          0 off | 1 auto | 2 wherever the geometry allows: 256x256 LDS-DMA forward / data-gradient core
  int bf16_v3_persist;   // EDRL_BF16_V3_PERSIST 0: one tile per workgroup (round 3) | 1 (default): persistent forward / plain data gradient | 2: also the epilogue variant
  int v3_epi_kmin;       // EDRL_V3_EPI_KMIN      the data gradient WITH the BatchNorm-backward epilogue takes the v3 core from K >= this (512: the K = 256 block-input gradients are all epilogue and run 1 ms per C2 step faster on three 128-row workgroups per CU)
  int v3_fwd_kmin;       // EDRL_V3_FWD_KMIN      the v3 forward takes K >= this (256 since the persistent form; 512 in round 3)
  int v3_stagger;        // EDRL_V3_STAGGER       1: staggered DMA issue of the two wave halves
  int bf16_v3s;          // EDRL_BF16_V3S         0 off | 1 auto | 2 wherever the geometry allows: 128x128 LDS-DMA core, two workgroups per CU (conv_bf16_v3s.hip)
  int v3s_kmin;          // EDRL_V3S_KMIN         (auto) the small-tile core takes 128-output-channel layers with K >= this (1024: the 3x3 layers of stage 2)
  int bf16_wgrad_v3;     // EDRL_BF16_WGRAD_V3    0 | 1 | 2: 256x256 LDS-DMA weight-gradient core
  int bf16_c64;          // EDRL_BF16_C64         0 | 1 | 2: weight-stationary 64 -> 64 3x3 kernel
  int bf16_k64;          // EDRL_BF16_K64         0 | 1 | 2: streaming expanding-1x1 kernel
  int bf16_k64_bwd;      // EDRL_BF16_K64_BWD     0: the expanding 1x1 layers' backward as two kernels (weight gradient, data gradient)
  int stem_pool_v8;      // EDRL_STEM_POOL_V8     0: 4-channel lanes in the fused stem pool kernels
  int bf16_epi_vw4;      // EDRL_BF16_EPI_VW4     1: 4-channel epilogue lanes in the 128-row kernel
  int bf16_ktail;        // EDRL_BF16_KTAIL       1: keep the load round past the last K tile
  int bf16_fwd_occ4;     // EDRL_BF16_FWD_OCC4    0: fused-operand forward back at 3 workgroups per CU
  int bf16_plain_occ4;   // EDRL_BF16_PLAIN_OCC4  0: plain forward back at 3 workgroups per CU
  int bf16_wgrad_bm64;   // EDRL_BF16_WGRAD_BM64  0: 128-row tiles for Co = 64 weight gradients
  long wgrad_target_bf16;// EDRL_WGRAD_TARGET_BF16 split-K workgroup target of the 128x128 bf16 weight gradient (1536)
  // fp32 kernels
  int gather_buf;        // EDRL_GATHER_BUF       0: raw-pointer operand loads instead of buffer descriptors
  int wgrad_fast;        // EDRL_WGRAD_FAST       0: generic row decode in the weight gradient
  long wgrad_target;     // EDRL_WGRAD_TARGET     split-K workgroup target override (0 = automatic)
  int narrow_below;      // EDRL_NARROW_BELOW     grids below this many workgroups use 64-wide N tiles (512)
  int gather_tail_split; // EDRL_GATHER_TAIL_SPLIT  fp32 gather: K-split of the tiles of a last partial quantum (<= 128 of 256 workgroups) + fix-up launch; 2: also of grids of <= 128 workgroups (2)
  int linear_smallm;     // EDRL_LINEAR_SMALLM    0: rows <= 64 Linear layers on the 128-row tiles
  int gather_variant;    // EDRL_GATHER_VARIANT   0 | 1 | 3 | 5: workgroups per CU of the gather kernel (1 = automatic)
  // diagnostic kernels: compiled only with -DEDRL_DIAG (make diag -> libedrl_hip_diag.so, never the shipped library)
  int diag_v3;           // EDRL_V3_DBG  1..4
  int diag_c64;          // EDRL_C64_DBG 1..2
};

const EdrlConfig& edrl_cfg();
