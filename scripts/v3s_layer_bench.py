"""Per-layer A/B of the bf16 forward / data-gradient kernels on the ResNet-50 layer shapes, interleaved in one process: the kernel
set without the small-tile core (EDRL_BF16_V3S=0: 256x256 core / 128-row kernel / weight-stationary kernels as dispatched) against
the small-tile LDS-DMA core forced (EDRL_BF16_V3S=2, conv_bf16_v3s.hip).  Plain forward with BatchNorm partials, plain data gradient,
data gradient with the BatchNorm-backward epilogue (sign bytes; accumulate where the layer is a block-input gradient).
usage: python scripts/v3s_layer_bench.py [images]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edrl_amd
ops = edrl_amd.ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device("cuda:0")
# name, Ci, H, Co, k, s, p
LAYERS = [("l1 1x1 256-64", 256, 56, 64, 1, 1, 0), ("l2 1x1 256-128", 256, 56, 128, 1, 1, 0), ("l2 3x3s2 128", 128, 56, 128, 3, 2, 1),
          ("l2 1x1 128-512", 128, 28, 512, 1, 1, 0), ("l2 1x1 512-128", 512, 28, 128, 1, 1, 0), ("l2 3x3 128", 128, 28, 128, 3, 1, 1),
          ("l2 ds 256-512 s2", 256, 56, 512, 1, 2, 0), ("l3 1x1 512-256", 512, 28, 256, 1, 1, 0), ("l3 3x3s2 256", 256, 28, 256, 3, 2, 1),
          ("l3 1x1 256-1024", 256, 14, 1024, 1, 1, 0), ("l3 1x1 1024-256", 1024, 14, 256, 1, 1, 0), ("l3 3x3 256", 256, 14, 256, 3, 1, 1),
          ("l3 ds 512-1024 s2", 512, 28, 1024, 1, 2, 0), ("l4 1x1 1024-512", 1024, 14, 512, 1, 1, 0), ("l4 3x3s2 512", 512, 14, 512, 3, 2, 1),
          ("l4 1x1 512-2048", 512, 7, 2048, 1, 1, 0), ("l4 1x1 2048-512", 2048, 7, 512, 1, 1, 0), ("l4 3x3 512", 512, 7, 512, 3, 1, 1),
          ("l4 ds 1024-2048 s2", 1024, 14, 2048, 1, 2, 0)]


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print(f"{N} images; ms per call: base = EDRL_BF16_V3S=0, v3s = EDRL_BF16_V3S=2; '-' = the small-tile core cannot take the geometry")
print(f"{'layer':20s} | {'fwd base':>8s} {'v3s':>7s} | {'dgrad base':>10s} {'v3s':>7s} | {'dgE base':>8s} {'v3s':>7s} | {'dgEA base':>9s} {'v3s':>7s} | max |diff| fwd / dgrad (bf16 ulps of the range)")
tot = {k: [0.0, 0.0] for k in ("fwd", "dg", "dge", "dgea")}
for name, Ci, H, Co, k, s, p in LAYERS:
    Ho = (H + 2 * p - k) // s + 1
    x = torch.randn(N, H, H, Ci, device=dev).bfloat16()
    w = (torch.randn(Co, k, k, Ci, device=dev) * 0.05).bfloat16()
    dy = torch.randn(N, Ho, Ho, Co, device=dev).bfloat16()
    wt = ops.permute_weight_bf16(w.float())
    fc = torch.randn(5, Ci, device=dev) * 0.1
    mask = torch.randint(0, 16, (N * H * H, Ci // 4), device=dev, dtype=torch.uint8)
    dxacc = torch.randn(N, H, H, Ci, device=dev).bfloat16()
    res, t = {}, {}
    for mode in ("0", "2"):
        edrl_amd._lib.set_switches(EDRL_BF16_V3S=mode)
        f = lambda: ops.conv2d_fwd_bf16(x, w, s, p, stats=True)
        d = lambda: ops.conv2d_dgrad_bf16(dy, wt, (N, H, H, Ci), s, p)
        de = lambda: ops.conv2d_dgrad_bn_bf16(dy, None, None, wt, (N, H, H, Ci), s, p, ep=(x, mask, fc, True))
        dea = lambda: ops.conv2d_dgrad_bn_bf16(dy, None, None, wt, (N, H, H, Ci), s, p, out=dxacc, accumulate=True, ep=(x, mask, fc, True))
        t[mode] = (timeit(f), timeit(d), timeit(de), timeit(dea))
        res[mode] = (f()[0].float(), d().float())
    edrl_amd._lib.set_switches(EDRL_BF16_V3S=None)
    df = float((res["0"][0] - res["2"][0]).abs().max() / res["0"][0].abs().max()) * 256
    dd = float((res["0"][1] - res["2"][1]).abs().max() / res["0"][1].abs().max()) * 256
    a, b = t["0"], t["2"]
    for key, i in (("fwd", 0), ("dg", 1), ("dge", 2), ("dgea", 3)):
        tot[key][0] += a[i]; tot[key][1] += min(a[i], b[i])
    print(f"{name:20s} | {a[0]:8.3f} {b[0]:7.3f} | {a[1]:10.3f} {b[1]:7.3f} | {a[2]:8.3f} {b[2]:7.3f} | {a[3]:9.3f} {b[3]:7.3f} | {df:.2f} / {dd:.2f}", flush=True)
    del x, w, dy, wt, mask, dxacc
print("sum over the layers (one call each), base -> best of (base, v3s): " + ", ".join(f"{k} {v[0]:.2f} -> {v[1]:.2f}" for k, v in tot.items()))
