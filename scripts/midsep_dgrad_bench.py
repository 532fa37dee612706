"""Data gradient of the 3x3 layers as the fp32 trunk's mid_sep units run it: plain operand (a materialised d_raw), epilogue that masks
with the sign bytes of the BatchNorm below and emits its partial sums -- against the plain data gradient (diagnostic, GPU only).
usage: python scripts/midsep_dgrad_bench.py [images]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edrl_amd
ops = edrl_amd.ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1056
dev = torch.device("cuda:0")


def timeit(fn, reps=8):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"{'layer':14s} | plain dgrad ms  TF/s | + mask epilogue ms  TF/s")
for name, C, H in [("l1 3x3 64", 64, 56), ("l2 3x3 128", 128, 28), ("l3 3x3 256", 256, 14), ("l4 3x3 512", 512, 7)]:
    d = torch.randn(N, H, H, C, device=dev)
    w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    wt = ops.permute_weight(w)
    x = torch.randn(N, H, H, C, device=dev)
    fc = torch.randn(5, C, device=dev) * 0.1
    fc[1] += 1.0; fc[2] += 1.0
    kb = torch.randint(0, 16, (N * H * H, C // 4), device=dev, dtype=torch.uint8)
    dx = torch.empty_like(x)
    flop = 2.0 * N * H * H * C * 9 * C
    t0 = timeit(lambda: ops.conv2d_dgrad(d, wt, tuple(x.shape), 1, 1, out=dx))
    t1 = timeit(lambda: ops.conv2d_dgrad_bn(d, None, None, wt, tuple(x.shape), 1, 1, ep=(x, kb, fc, True)))
    print(f"{name:14s} | {t0:8.3f} {flop / t0 / 1e9:7.1f} | {t1:8.3f} {flop / t1 / 1e9:7.1f}")
