"""Per-layer timing of the MFMA conv kernels on the ResNet-50 layer shapes (diagnostic, GPU only).
usage: python scripts/conv_layer_bench.py [images]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edrl_amd
ops = edrl_amd.ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
# (name, Ci, H, Co, k, s, p, count in R50)
L = [("stem7x7", 4, 224, 64, 7, 2, 3, 1),
     ("l1 1x1 64-64", 64, 56, 64, 1, 1, 0, 1), ("l1 3x3 64", 64, 56, 64, 3, 1, 1, 3), ("l1 1x1 64-256", 64, 56, 256, 1, 1, 0, 4),
     ("l1 1x1 256-64", 256, 56, 64, 1, 1, 0, 2),
     ("l2 1x1 256-128", 256, 56, 128, 1, 1, 0, 1), ("l2 3x3s2 128", 128, 56, 128, 3, 2, 1, 1), ("l2 1x1 128-512", 128, 28, 512, 1, 1, 0, 4),
     ("l2 1x1 512-128", 512, 28, 128, 1, 1, 0, 3), ("l2 3x3 128", 128, 28, 128, 3, 1, 1, 3), ("l2 ds 256-512 s2", 256, 56, 512, 1, 2, 0, 1),
     ("l3 1x1 512-256", 512, 28, 256, 1, 1, 0, 1), ("l3 3x3s2 256", 256, 28, 256, 3, 2, 1, 1), ("l3 1x1 256-1024", 256, 14, 1024, 1, 1, 0, 6),
     ("l3 1x1 1024-256", 1024, 14, 256, 1, 1, 0, 5), ("l3 3x3 256", 256, 14, 256, 3, 1, 1, 5), ("l3 ds 512-1024 s2", 512, 28, 1024, 1, 2, 0, 1),
     ("l4 1x1 1024-512", 1024, 14, 512, 1, 1, 0, 1), ("l4 3x3s2 512", 512, 14, 512, 3, 2, 1, 1), ("l4 1x1 512-2048", 512, 7, 2048, 1, 1, 0, 3),
     ("l4 1x1 2048-512", 2048, 7, 512, 1, 1, 0, 2), ("l4 3x3 512", 512, 7, 512, 3, 1, 1, 2), ("l4 ds 1024-2048 s2", 1024, 14, 2048, 1, 2, 0, 1)]


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot = {"fwd": 0, "dgrad": 0, "wgrad": 0, "flop": 0}
print(f"{'layer':22s} {'GFLOP':>8s} | {'fwd ms':>8s} {'TF':>6s} | {'dgrad ms':>8s} {'TF':>6s} | {'wgrad ms':>8s} {'TF':>6s} | minbytes-ms@5TB/s")
for name, Ci, H, Co, k, s, p, cnt in L:
    Ho = (H + 2 * p - k) // s + 1
    x = torch.randn(N, H, H, Ci, device=dev)
    w = torch.randn(Co, k, k, Ci, device=dev) * 0.05
    dy = torch.randn(N, Ho, Ho, Co, device=dev)
    wt = ops.permute_weight(w)
    flop = 2.0 * N * Ho * Ho * Co * k * k * Ci
    y = torch.empty(N, Ho, Ho, Co, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
    if os.environ.get("BENCH_FWD_STATS") == "1" and Ci % 16 == 0:     # forward with the fused BatchNorm partials (as in the step)
        shift = torch.zeros(Co, device=dev)
        tf = timeit(lambda: ops.conv2d_fwd_stats(x, w, shift, s, p))
    else:
        tf = timeit(lambda: ops.conv2d_fwd(x, w, stride=s, pad=p, out=y))
    td = timeit(lambda: ops.conv2d_dgrad(dy, wt, tuple(x.shape), s, p, out=dx)) if Ci % 4 == 0 and name != "stem7x7" else float("nan")
    tw = timeit(lambda: ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, out=dw))
    mb = (x.numel() + y.numel()) * 4 / 5e12 * 1e3
    print(f"{name:22s} {flop/1e9:8.1f} | {tf:8.3f} {flop/tf/1e9:6.1f} | {td:8.3f} {flop/td/1e9:6.1f} | {tw:8.3f} {flop/tw/1e9:6.1f} | {mb:6.3f}  x{cnt}")
    tot["fwd"] += tf * cnt; tot["wgrad"] += tw * cnt; tot["flop"] += flop * cnt
    if td == td:
        tot["dgrad"] += td * cnt
    del x, w, dy, wt, y, dx, dw
print(f"R50 total per {N} images: fwd {tot['fwd']:.1f} ms ({tot['flop']/tot['fwd']/1e9:.1f} TF)  dgrad {tot['dgrad']:.1f} ms  wgrad {tot['wgrad']:.1f} ms ({tot['flop']/tot['wgrad']/1e9:.1f} TF)")
