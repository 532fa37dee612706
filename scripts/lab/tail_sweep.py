"""How much do partial last rounds cost the fp32 implicit-GEMM kernel?  Forward of two 3x3 layers over a sweep of image counts
(diagnostic, GPU): TFLOP/s against the number of 128x128 workgroups / 1024 resident slots.
usage: python scripts/tail_sweep.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edrl_amd
ops = edrl_amd.ops
dev = torch.device("cuda:0")


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, C, H in (("l3 3x3 256", 256, 14), ("l4 3x3 512", 512, 7)):
    w = torch.randn(C, 3, 3, C, device=dev) * 0.05
    for N in (669, 836, 1003, 1024, 1056, 1100, 1170, 1254, 1337, 1338, 1420, 1504, 1672):
        x = torch.randn(N, H, H, C, device=dev)
        t = min(timeit(lambda: ops.conv2d_fwd(x, w, stride=1, pad=1)) for _ in range(3))
        wgs = -(-N * H * H // 128) * (C // 128)
        fl = 2.0 * N * H * H * C * 9 * C
        print(f"{name}  N {N:5d}  wgs {wgs:6d} = {wgs / 1024:5.2f} rounds  {t:7.3f} ms  {fl / t / 1e9:6.1f} TFLOP/s", flush=True)
