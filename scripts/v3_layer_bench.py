"""A/B of the two bf16 implicit-GEMM cores on the ResNet-50 layer shapes (GPU, diagnostic): the 128-row register-staged kernel
(EDRL_BF16_V3=0) against the 256x256 LDS-DMA core (EDRL_BF16_V3=2), forward and data gradient, interleaved rounds in ONE
process; also checks that both produce the same tensor (bf16 outputs: max |diff| relative to max |ref|, expected <= 1 bf16 ulp
of accumulation-order noise) and that the fused BatchNorm chunk partials agree.
usage: python scripts/v3_layer_bench.py [images] [layer-substring]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edrl_amd
ops = edrl_amd.ops
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2112
only = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
L = [("l2 3x3s2 128", 128, 56, 128, 3, 2, 1, 1), ("l2 1x1 128-512", 128, 28, 512, 1, 1, 0, 4),
     ("l2 1x1 512-128", 512, 28, 128, 1, 1, 0, 3), ("l2 3x3 128", 128, 28, 128, 3, 1, 1, 3), ("l2 ds 256-512 s2", 256, 56, 512, 1, 2, 0, 1),
     ("l3 1x1 512-256", 512, 28, 256, 1, 1, 0, 1), ("l3 3x3s2 256", 256, 28, 256, 3, 2, 1, 1), ("l3 1x1 256-1024", 256, 14, 1024, 1, 1, 0, 6),
     ("l3 1x1 1024-256", 1024, 14, 256, 1, 1, 0, 5), ("l3 3x3 256", 256, 14, 256, 3, 1, 1, 5), ("l3 ds 512-1024 s2", 512, 28, 1024, 1, 2, 0, 1),
     ("l4 1x1 1024-512", 1024, 14, 512, 1, 1, 0, 1), ("l4 3x3s2 512", 512, 14, 512, 3, 2, 1, 1), ("l4 1x1 512-2048", 512, 7, 2048, 1, 1, 0, 3),
     ("l4 1x1 2048-512", 2048, 7, 512, 1, 1, 0, 2), ("l4 3x3 512", 512, 7, 512, 3, 1, 1, 2), ("l4 ds 1024-2048 s2", 1024, 14, 2048, 1, 2, 0, 1)]


def timeit(fn, reps=3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# third column (round 4): the persistent form of the v3 core (conv_bf16_v3p.hip; round 3 listed the staggered variant there)
MODES = {"old": {"EDRL_BF16_V3": "0", "EDRL_BF16_V3_PERSIST": "0"}, "v3": {"EDRL_BF16_V3": "2", "EDRL_BF16_V3_PERSIST": "0"},
         "v3stag": {"EDRL_BF16_V3": "2", "EDRL_BF16_V3_PERSIST": "1"}}


def setmode(m):
    os.environ.update(MODES[m]); edrl_amd._lib.set_switches()      # the library reads its environment once: re-read


def ab(fn, rounds=3):
    best = {m: 1e9 for m in MODES}
    for r in range(rounds + 1):
        for m in MODES:
            setmode(m)
            t = timeit(fn, 1 if r == 0 else 3)
            if r:
                best[m] = min(best[m], t)
    return best


def run(mode, fn):
    setmode(mode)
    o = fn()
    torch.cuda.synchronize()
    return o


print(f"{'layer':20s} {'GFLOP':>7s} | fwd ms (TF): 128-row kernel, v3 (one tile per workgroup), v3 persistent | dgrad ms (TF): same three | max rel diff vs the 128-row kernel: fwd v3/stag, stats, dgrad v3/stag")
tot = {}
for name, Ci, H, Co, k, s, p, cnt in L:
    if only and only not in name:
        continue
    Ho = (H + 2 * p - k) // s + 1
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn(N, H, H, Ci, device=dev, generator=g).bfloat16()
    w = torch.randn(Co, k, k, Ci, device=dev, generator=g) * 0.05
    wb = w.bfloat16(); wt = ops.permute_weight_bf16(w)
    dy = torch.randn(N, Ho, Ho, Co, device=dev, generator=g).bfloat16()
    flop = 2.0 * N * Ho * Ho * Co * k * k * Ci
    dx = torch.empty_like(x)
    f = lambda: ops.conv2d_fwd_bf16(x, wb, s, p, stats=True)
    d = lambda: ops.conv2d_dgrad_bf16(dy, wt, tuple(x.shape), s, p, out=dx)
    rel = lambda a, b: float((a.float() - b.float()).abs().max() / b.float().abs().max())
    y0, st0, _ = run("old", f); y16, st16, _ = run("v3", f); y32, st32, _ = run("v3stag", f)
    run("old", d); d0 = dx.clone(); run("v3", d); d16 = dx.clone(); run("v3stag", d)
    errs = f"{rel(y16, y0):.1e}/{rel(y32, y0):.1e}, {rel(st32[:, :2], st0[:, :2]):.1e}, {rel(d16, d0):.1e}/{rel(dx, d0):.1e}"
    tf = ab(f); td = ab(d)
    T = lambda t: flop / t / 1e9
    print(f"{name:20s} {flop/1e9:7.1f} | " + " ".join(f"{tf[m]:6.3f} ({T(tf[m]):4.0f})" for m in MODES) + " | " +
          " ".join(f"{td[m]:6.3f} ({T(td[m]):4.0f})" for m in MODES) + f" | {errs}  x{cnt}", flush=True)
    for m in MODES:
        tot["f" + m] = tot.get("f" + m, 0.0) + tf[m] * cnt
        tot["d" + m] = tot.get("d" + m, 0.0) + td[m] * cnt
print(f"sum over the listed layers x count, {N} images: fwd " + " / ".join(f"{tot['f' + m]:.2f}" for m in MODES) + " ms, dgrad " +
      " / ".join(f"{tot['d' + m]:.2f}" for m in MODES) + " ms  (" + ", ".join(MODES) + ")")
