"""Timing of the 64->64 3x3 kernel's masked-gradient (EPI 1) form against the plain data gradient (diagnostic)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import edrl_amd
ops = edrl_amd.ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
x = torch.randn(N, 56, 56, 64, device=dev).bfloat16()
w = (torch.randn(64, 3, 3, 64, device=dev) * 0.05).bfloat16()
wt = ops.permute_weight_bf16(w.float())
xr = torch.randn(N, 56, 56, 64, device=dev).bfloat16()
mask = torch.randint(0, 16, (N * 56 * 56, 16), device=dev, dtype=torch.uint8)
fc = torch.ones(5, 64, device=dev)
dx = torch.empty_like(x)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for mode in ("0", "1"):
    edrl_amd._lib.set_switches(EDRL_BF16_C64=mode)
    print("C64", mode, "dgrad %.3f ms" % t(lambda: ops.conv2d_dgrad_bf16(x, wt, tuple(x.shape), 1, 1, out=dx)),
          "dgrad+epilogue %.3f ms" % t(lambda: ops.conv2d_dgrad_bn_bf16(x, None, None, wt, tuple(x.shape), 1, 1, out=dx, ep=(xr, mask, fc, True))), flush=True)
