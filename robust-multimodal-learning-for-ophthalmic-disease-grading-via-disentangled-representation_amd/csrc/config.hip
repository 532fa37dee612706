// The library's run-time switches (edrl_config.h): one read of the environment, re-read on request.
#include "edrl_config.h"
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <mutex>

static EdrlConfig g_cfg;
static std::once_flag g_once;

static int env_int(const char* name, int dflt) { const char* e = getenv(name); return (e && *e) ? atoi(e) : dflt; }
static long env_long(const char* name, long dflt) { const char* e = getenv(name); return (e && *e) ? atol(e) : dflt; }

static void load(EdrlConfig& c) {
  c.bf16_v3 = env_int("EDRL_BF16_V3", 1);
  c.bf16_v3_persist = env_int("EDRL_BF16_V3_PERSIST", 1);
  c.v3_epi_kmin = env_int("EDRL_V3_EPI_KMIN", 512);
  c.v3_fwd_kmin = env_int("EDRL_V3_FWD_KMIN", 256);
  c.v3_stagger = env_int("EDRL_V3_STAGGER", 0);
  c.bf16_v3s = env_int("EDRL_BF16_V3S", 1);
  c.v3s_kmin = env_int("EDRL_V3S_KMIN", 1024);
  c.bf16_wgrad_v3 = env_int("EDRL_BF16_WGRAD_V3", 1);
  c.bf16_c64 = env_int("EDRL_BF16_C64", 1);
  c.bf16_k64 = env_int("EDRL_BF16_K64", 1);
  c.bf16_k64_bwd = env_int("EDRL_BF16_K64_BWD", 1);
  c.stem_pool_v8 = env_int("EDRL_STEM_POOL_V8", 1);
  c.bf16_epi_vw4 = env_int("EDRL_BF16_EPI_VW4", 0);
  c.bf16_ktail = env_int("EDRL_BF16_KTAIL", 0);
  c.bf16_fwd_occ4 = env_int("EDRL_BF16_FWD_OCC4", 1);
  c.bf16_plain_occ4 = env_int("EDRL_BF16_PLAIN_OCC4", 1);
  c.bf16_wgrad_bm64 = env_int("EDRL_BF16_WGRAD_BM64", 1);
  c.wgrad_target_bf16 = env_long("EDRL_WGRAD_TARGET_BF16", 1536L);
  c.gather_buf = env_int("EDRL_GATHER_BUF", 1);
  c.wgrad_fast = env_int("EDRL_WGRAD_FAST", 1);
  c.wgrad_target = env_long("EDRL_WGRAD_TARGET", 0L);
  c.narrow_below = env_int("EDRL_NARROW_BELOW", 512);
  c.gather_tail_split = env_int("EDRL_GATHER_TAIL_SPLIT", 2);
  c.linear_smallm = env_int("EDRL_LINEAR_SMALLM", 1);
  c.gather_variant = env_int("EDRL_GATHER_VARIANT", 1);
  c.diag_v3 = env_int("EDRL_V3_DBG", 0);
  c.diag_c64 = env_int("EDRL_C64_DBG", 0);
}

const EdrlConfig& edrl_cfg() {
  std::call_once(g_once, []() { load(g_cfg); });
  return g_cfg;
}

// C-ABI (include/edrl_hip.h): re-read the EDRL_* environment switches.  Not to be called while launches are in flight on
// other threads (tests and A/B scripts call it between launches).  Returns 1 when the library holds diagnostic kernels (-DEDRL_DIAG).
extern "C" int edrl_config_reload(void) {
  (void)edrl_cfg();
  load(g_cfg);
#ifdef EDRL_DIAG
  return 1;
#else
  return 0;
#endif
}

// C-ABI (include/edrl_hip.h): an empty one-wave dispatch.  The call tracer (_lib.call with tracing on, scripts/step_trace.py)
// issues one in front of every library call, so that the dispatches of a rocprofv3 kernel trace / --pmc pass can be cut into
// calls by position: the k-th `edrl_trace_mark_kernel` row precedes the kernels of the k-th logged call.  Reads and writes nothing.
__global__ void edrl_trace_mark_kernel() {}
extern "C" int edrl_trace_mark(hipStream_t stream) {
  hipLaunchKernelGGL(edrl_trace_mark_kernel, dim3(1), dim3(64), 0, stream);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
