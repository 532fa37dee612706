// Launch interface of the 256x256 LDS-DMA bf16 implicit-GEMM core (conv_bf16_v3.hip), used by the C-ABI launchers in conv_bf16.hip.
#pragma once
#include <hip/hip_runtime.h>
#include "conv_geom.h"

// true when the geometry can run on the v3 core (channel counts, descriptor footprints) AND is expected to be faster there.
bool gather_bf16_v3_ok(const GatherGeom& g, bool dgrad);
// forward (dgrad = false) or data gradient (one parity class per call) of the plain bf16 convolution; honours GF_STATS / GF_ACCUM.
int launch_gather_bf16_v3(const void* src, const void* wm, void* dst, const GatherGeom& g, bool dgrad, hipStream_t st);
