"""Timing of the 64->64 3x3 kernel's diagnostic builds (EDRL_C64_DBG with EDRL_ALLOW_DIAGNOSTIC_KERNELS=1; wrong outputs by construction)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import edrl_amd
ops = edrl_amd.ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2112
x = torch.randn(N, 56, 56, 64, device=dev).bfloat16()
w = (torch.randn(64, 3, 3, 64, device=dev) * 0.05).bfloat16()
wt = ops.permute_weight_bf16(w.float())
dx = torch.empty_like(x)
# needs the diagnostic build: `make -C <package>/csrc diag` and EDRL_LIB_PATH=<package>/libedrl_hip_diag.so (the shipped library holds no
# diagnostic kernels)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for dbg in ("0", "1", "2"):
    assert edrl_amd._lib.set_switches(EDRL_C64_DBG=dbg) == 1, "load libedrl_hip_diag.so (EDRL_LIB_PATH)"
    print("dbg", dbg, "fwd+stats %.3f ms" % t(lambda: ops.conv2d_fwd_bf16(x, w, 1, 1, stats=True)),
          "fwd %.3f ms" % t(lambda: ops.conv2d_fwd_bf16(x, w, 1, 1)), "dgrad %.3f ms" % t(lambda: ops.conv2d_dgrad_bf16(x, wt, tuple(x.shape), 1, 1, out=dx)), flush=True)
