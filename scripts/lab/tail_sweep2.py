"""128x128 against 128x64 tiles of the fp32 implicit-GEMM kernel at exact and inexact multiples of 256 workgroups (diagnostic, GPU)."""
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


for name, C, H, k in (("l3 3x3 256", 256, 14, 3), ("l4 3x3 512", 512, 7, 3), ("l4 1x1 2048-512", 2048, 7, 1), ("l3 1x1 1024-256", 1024, 14, 1)):
    Co = C if k == 3 else C // 4
    w = torch.randn(Co, k, k, C, device=dev) * 0.05
    for N in (1003, 1024):
        x = torch.randn(N, H, H, C, device=dev)
        fl = 2.0 * N * H * H * C * k * k * Co
        for nb in ("768", "1000000"):
            edrl_amd._lib.set_switches(EDRL_NARROW_BELOW=nb)
            t = min(timeit(lambda: ops.conv2d_fwd(x, w, stride=1, pad=k // 2)) for _ in range(3))
            bn = 64 if nb != "768" else 128
            wgs = -(-N * H * H // 128) * (Co // bn)
            print(f"{name}  N {N:5d}  BN {bn:3d}  wgs {wgs:6d} = {wgs / 256:6.2f} x 256  {t:7.3f} ms  {fl / t / 1e9:6.1f} TFLOP/s", flush=True)
