"""A/B of the two bf16 weight-gradient cores on the wide ResNet-50 layer shapes (GPU, diagnostic): the 128x128 register-staged
kernel (EDRL_BF16_WGRAD_V3=0) against the 256x256 LDS-DMA core (EDRL_BF16_WGRAD_V3=2), interleaved rounds in ONE process; also
prints the max difference between the two results relative to the result's max (fp32 outputs, different split plans).
usage: python scripts/wgrad_layer_bench.py [images] [layer-substring]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edrl_amd
ops = edrl_amd.ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2112
only = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
L = [("l3 1x1 512-256", 512, 28, 256, 1, 1, 0, 1), ("l3 3x3s2 256", 256, 28, 256, 3, 2, 1, 1), ("l3 1x1 256-1024", 256, 14, 1024, 1, 1, 0, 6),
     ("l3 1x1 1024-256", 1024, 14, 256, 1, 1, 0, 5), ("l3 3x3 256", 256, 14, 256, 3, 1, 1, 5), ("l3 ds 512-1024 s2", 512, 28, 1024, 1, 2, 0, 1),
     ("l4 1x1 1024-512", 1024, 14, 512, 1, 1, 0, 1), ("l4 3x3s2 512", 512, 14, 512, 3, 2, 1, 1), ("l4 1x1 512-2048", 512, 7, 2048, 1, 1, 0, 3),
     ("l4 1x1 2048-512", 2048, 7, 512, 1, 1, 0, 2), ("l4 3x3 512", 512, 7, 512, 3, 1, 1, 2), ("l4 ds 1024-2048 s2", 1024, 14, 2048, 1, 2, 0, 1)]
MODES = {"old": "0", "v3": "2"}


def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def ab(fn, rounds=3):
    best = {m: 1e9 for m in MODES}
    for r in range(rounds + 1):
        for m in MODES:
            edrl_amd._lib.set_switches(EDRL_BF16_WGRAD_V3=MODES[m])
            t = timeit(fn, 1 if r == 0 else 3)
            if r:
                best[m] = min(best[m], t)
    return best


print(f"{'layer':20s} {'GFLOP':>7s} | wgrad ms (TFLOP/s): 128x128 kernel, v3 | max rel diff", flush=True)
tot = {m: 0.0 for m in MODES}
for name, Ci, H, Co, k, s, p, cnt in L:
    if only and only not in name:
        continue
    Ho = (H + 2 * p - k) // s + 1
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, H, H, Ci, device=dev, generator=g).bfloat16()
    dy = torch.randn(N, Ho, Ho, Co, device=dev, generator=g).bfloat16()
    flop = 2.0 * N * Ho * Ho * Co * k * k * Ci
    dw = torch.empty(Co, k, k, Ci, device=dev)
    f = lambda: ops.conv2d_wgrad_bf16(dy, x, (Co, k, k, Ci), s, p, out=dw)
    edrl_amd._lib.set_switches(EDRL_BF16_WGRAD_V3="0"); f(); torch.cuda.synchronize(); d0 = dw.clone()
    edrl_amd._lib.set_switches(EDRL_BF16_WGRAD_V3="2"); f(); torch.cuda.synchronize()
    err = float((dw - d0).abs().max() / d0.abs().max())
    t = ab(f)
    print(f"{name:20s} {flop/1e9:7.1f} | " + " ".join(f"{t[m]:6.3f} ({flop / t[m] / 1e9:4.0f})" for m in MODES) + f" | {err:.1e}  x{cnt}", flush=True)
    for m in MODES:
        tot[m] += t[m] * cnt
print(f"sum over the listed layers x count, {N} images: " + " / ".join(f"{tot[m]:.2f}" for m in MODES) + " ms  (" + ", ".join(MODES) + ")")
