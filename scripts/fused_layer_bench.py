"""Per-layer timing of the fused-BatchNorm conv kernels against the separate-pass kernels they replace, on the ResNet-50
layer shapes (diagnostic, GPU only).  usage: python scripts/fused_layer_bench.py [images] [layer-substring]

columns (ms):  fwd   = plain conv (+stats epilogue)        | fwd+bn = conv that applies BN+ReLU to its input in the operand load
               bnap  = the bn_apply pass on the input tensor that fwd+bn makes unnecessary
               dg    = plain data gradient                   | dgF = d_raw formed in the operand load | dgFE = dgF + masked
                       gradient + BN-backward partial sums from the epilogue (ReLU decision recomputed: gradients inside a block)
                       | dgFA = dgFE with sign bytes, accumulating into dx (the block-input gradient)
               bnbw  = the colstat + bn_bwd_apply passes (on the conv-output-sized tensor) that dgF/wgF make unnecessary
               wg    = plain weight gradient                 | wgF = d_raw in the dY load | wgFX = wgF + BN+ReLU in the X load
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import edrl_amd
from edrl_amd import _lib as LL
ops = edrl_amd.ops
P = LL.ptr

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1056
only = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda:0")
LAYERS = [("l1 1x1 64-64", 64, 56, 64, 1, 1, 0, 1), ("l1 3x3 64", 64, 56, 64, 3, 1, 1, 3), ("l1 1x1 64-256", 64, 56, 256, 1, 1, 0, 4),
          ("l1 1x1 256-64", 256, 56, 64, 1, 1, 0, 2),
          ("l2 1x1 256-128", 256, 56, 128, 1, 1, 0, 1), ("l2 3x3s2 128", 128, 56, 128, 3, 2, 1, 1), ("l2 1x1 128-512", 128, 28, 512, 1, 1, 0, 4),
          ("l2 1x1 512-128", 512, 28, 128, 1, 1, 0, 3), ("l2 3x3 128", 128, 28, 128, 3, 1, 1, 3), ("l2 ds 256-512 s2", 256, 56, 512, 1, 2, 0, 1),
          ("l3 1x1 512-256", 512, 28, 256, 1, 1, 0, 1), ("l3 3x3s2 256", 256, 28, 256, 3, 2, 1, 1), ("l3 1x1 256-1024", 256, 14, 1024, 1, 1, 0, 6),
          ("l3 1x1 1024-256", 1024, 14, 256, 1, 1, 0, 5), ("l3 3x3 256", 256, 14, 256, 3, 1, 1, 5), ("l3 ds 512-1024 s2", 512, 28, 1024, 1, 2, 0, 1),
          ("l4 1x1 1024-512", 1024, 14, 512, 1, 1, 0, 1), ("l4 3x3s2 512", 512, 14, 512, 3, 2, 1, 1), ("l4 1x1 512-2048", 512, 7, 2048, 1, 1, 0, 3),
          ("l4 1x1 2048-512", 2048, 7, 512, 1, 1, 0, 2), ("l4 3x3 512", 512, 7, 512, 3, 1, 1, 2), ("l4 ds 1024-2048 s2", 1024, 14, 2048, 1, 2, 0, 1)]


def timeit(fn, reps=4):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def fcoef(C):
    fc = torch.empty(5, C, device=dev)
    fc[0] = 0.1 * torch.randn(C, device=dev); fc[1] = 1.0 + 0.1 * torch.rand(C, device=dev)
    fc[2] = fc[1] * (0.5 + torch.rand(C, device=dev)); fc[3] = 0.1 * torch.randn(C, device=dev)
    fc[4] = fc[3] - fc[0] * fc[2]
    return fc


def bcoef(C):
    bc = torch.empty(4, C, device=dev)
    bc[0] = 0.5 + torch.rand(C, device=dev); bc[1] = 0.01 * torch.randn(C, device=dev)
    bc[2] = 0.01 * torch.randn(C, device=dev); bc[3] = 0.1 * torch.randn(C, device=dev)
    return bc


hdr = f"{'layer':20s} {'GFLOP':>7s} | {'fwd':>6s} {'fwd+bn':>6s} {'bnap':>6s} | {'dg':>6s} {'dgF':>6s} {'dgFE':>6s} {'dgFA':>6s} {'bnbw':>6s} | {'wg':>6s} {'wgF':>6s} {'wgFX':>6s} | sep -> fused (ms, x count)"
print(hdr)
tot_sep = tot_fus = 0.0
for name, Ci, H, Co, k, s, p, cnt in LAYERS:
    if only and only not in name:
        continue
    Ho = (H + 2 * p - k) // s + 1
    x = torch.randn(N, H, H, Ci, device=dev)
    w = torch.randn(Co, k, k, Ci, device=dev) * 0.05
    dy = torch.randn(N, Ho, Ho, Co, device=dev)
    yraw = torch.randn(N, Ho, Ho, Co, device=dev)
    wt = ops.permute_weight(w)
    flop = 2.0 * N * Ho * Ho * Co * k * k * Ci
    fin, fout, bout = fcoef(Ci), fcoef(Co), bcoef(Co)
    dx = torch.empty_like(x)
    M_in, M_out = N * H * H, N * Ho * Ho
    t_fwd = timeit(lambda: ops.conv2d_fwd_stats(x, w, None, s, p))
    t_fwdbn = timeit(lambda: ops.conv2d_fwd_bnin_stats(x, fin, w, s, p))
    act = torch.empty_like(x); mask = torch.empty(M_in, Ci // 4, device=dev, dtype=torch.uint8)
    t_bnap = timeit(lambda: LL.call("edrl_bn_apply_f32", P(x), P(fin[0]), P(fin[2]), P(fin[3]), None, P(act), P(mask), M_in, Ci, Ci, 1))
    del act
    t_dg = timeit(lambda: ops.conv2d_dgrad(dy, wt, tuple(x.shape), s, p, out=dx))
    t_dgF = timeit(lambda: ops.conv2d_dgrad_bn(dy, yraw, bout, wt, tuple(x.shape), s, p, out=dx))
    t_dgFE = timeit(lambda: ops.conv2d_dgrad_bn(dy, yraw, bout, wt, tuple(x.shape), s, p, out=dx, ep=(x, None, fin, True)))
    kbytes = torch.randint(0, 16, (M_in, Ci // 4), device=dev, dtype=torch.uint8)
    dx.zero_()
    t_dgFA = timeit(lambda: ops.conv2d_dgrad_bn(dy, yraw, bout, wt, tuple(x.shape), s, p, out=dx, accumulate=True,
                                                ep=(x, kbytes, fin, True)))
    del kbytes
    # separate BN backward on the conv-output-sized tensor
    d_raw = torch.empty_like(dy); dgm = torch.empty(Co, device=dev); dbt = torch.empty(Co, device=dev)
    gam = torch.ones(Co, device=dev); mk = torch.empty(M_out, Co // 4, device=dev, dtype=torch.uint8).fill_(0xf)
    nb = LL.query("edrl_bn_workspace_bytes", M_out, Co) + 2 * Co * 4
    ws = torch.empty(nb // 4, device=dev)
    t_bnbw = timeit(lambda: LL.call("edrl_bn_bwd_f32", P(dy), None, P(mk), P(yraw), P(fout[0]), P(fout[1]), P(gam), P(dgm), P(dbt), 0,
                                     P(d_raw), None, 0, M_out, Co, Co, P(ws), nb))
    del d_raw, ws, mk
    t_wg = timeit(lambda: ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p))
    t_wgF = timeit(lambda: ops.conv2d_wgrad_bn(dy, yraw, bout, x, None, tuple(w.shape), s, p))
    t_wgFX = timeit(lambda: ops.conv2d_wgrad_bn(dy, yraw, bout, x, fin, tuple(w.shape), s, p))
    sep = t_fwd + t_bnap + t_dg + t_bnbw + t_wg
    fus = t_fwdbn + t_dgFE + t_wgFX
    tot_sep += sep * cnt; tot_fus += fus * cnt
    print(f"{name:20s} {flop/1e9:7.1f} | {t_fwd:6.3f} {t_fwdbn:6.3f} {t_bnap:6.3f} | {t_dg:6.3f} {t_dgF:6.3f} {t_dgFE:6.3f} {t_dgFA:6.3f} {t_bnbw:6.3f} | "
          f"{t_wg:6.3f} {t_wgF:6.3f} {t_wgFX:6.3f} | {sep:6.3f} -> {fus:6.3f} x{cnt}", flush=True)
    del x, w, dy, yraw, wt, dx
print(f"R50 body per {N} images (one conv+BN unit = fwd + bn_apply(in) + dgrad + bn_bwd(out) + wgrad): separate {tot_sep:.1f} ms -> fused {tot_fus:.1f} ms")
