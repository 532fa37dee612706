import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, edrl_amd
ops = edrl_amd.ops
dev = torch.device("cuda:0")
N = 2112
def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
# needs the diagnostic build: `make -C <package>/csrc diag` and EDRL_LIB_PATH=<package>/libedrl_hip_diag.so (the shipped library holds no
# diagnostic kernels)
os.environ["EDRL_BF16_V3"] = "2"
for name, Ci, H, Co, k, s, p in [("l3 3x3 256", 256, 14, 256, 3, 1, 1), ("l4 3x3 512", 512, 7, 512, 3, 1, 1), ("l4 1x1 2048-512", 2048, 7, 512, 1, 1, 0), ("l3 1x1 1024-256", 1024, 14, 256, 1, 1, 0)]:
    x = torch.randn(N, H, H, Ci, device=dev).bfloat16()
    wb = (torch.randn(Co, k, k, Ci, device=dev) * 0.05).bfloat16()
    Ho = (H + 2 * p - k) // s + 1
    flop = 2.0 * N * Ho * Ho * Co * k * k * Ci
    r = []
    for dbg in ("0", "1", "2"):
        assert edrl_amd._lib.set_switches(EDRL_V3_DBG=dbg) == 1, "load libedrl_hip_diag.so (EDRL_LIB_PATH)"
        t = timeit(lambda: ops.conv2d_fwd_bf16(x, wb, s, p))
        r.append(f"dbg{dbg}: {t:.3f} ms {flop/t/1e9:6.0f} TF")
    tiles = ((N * Ho * Ho + 255) // 256) * (Co // 256)
    print(f"{name:18s} tiles {tiles} ({tiles/256:.2f} rounds) KT {k*k*Ci//64} | " + " | ".join(r), flush=True)
