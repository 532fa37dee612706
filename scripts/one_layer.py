"""Run one conv layer shape repeatedly (for rocprofv3 --pmc passes). usage: one_layer.py Ci H Co k s p [images] [mode]
modes: fwd dgrad wgrad (plain) | fwdst (fwd + stats epilogue) fwdbn (BN+ReLU operand load + stats) dgF dgFE wgF wgFX (fused-BatchNorm
variants, see scripts/fused_layer_bench.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, edrl_amd
ops = edrl_amd.ops
Ci, H, Co, k, s, p = map(int, sys.argv[1:7])
N = int(sys.argv[7]) if len(sys.argv) > 7 else 256
mode = sys.argv[8] if len(sys.argv) > 8 else "fwd"
reps = int(os.environ.get("REPS", "10"))
dev = torch.device("cuda:0")
Ho = (H + 2 * p - k) // s + 1
x = torch.randn(N, H, H, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev) * 0.05
dy = torch.randn(N, Ho, Ho, Co, device=dev); wt = ops.permute_weight(w)
yraw = torch.randn(N, Ho, Ho, Co, device=dev)
y = torch.empty(N, Ho, Ho, Co, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)


def coef(rows, C):
    t = torch.randn(rows, C, device=dev) * 0.1
    t[1] += 1.0; t[2] += 1.0
    return t


fin, bout = coef(5, Ci), coef(4, Co)
fn = {"fwd": lambda: ops.conv2d_fwd(x, w, stride=s, pad=p, out=y),
      "dgrad": lambda: ops.conv2d_dgrad(dy, wt, tuple(x.shape), s, p, out=dx),
      "wgrad": lambda: ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, out=dw),
      "fwdst": lambda: ops.conv2d_fwd_stats(x, w, None, s, p),
      "fwdbn": lambda: ops.conv2d_fwd_bnin_stats(x, fin, w, s, p),
      "dgF": lambda: ops.conv2d_dgrad_bn(dy, yraw, bout, wt, tuple(x.shape), s, p, out=dx),
      "dgFE": lambda: ops.conv2d_dgrad_bn(dy, yraw, bout, wt, tuple(x.shape), s, p, out=dx, ep=(x, None, fin, True)),
      "wgF": lambda: ops.conv2d_wgrad_bn(dy, yraw, bout, x, None, tuple(w.shape), s, p),
      "wgFX": lambda: ops.conv2d_wgrad_bn(dy, yraw, bout, x, fin, tuple(w.shape), s, p)}[mode]
for _ in range(reps):
    fn()
torch.cuda.synchronize()
print("done", 2.0 * N * Ho * Ho * Co * k * k * Ci / 1e9, "GFLOP per launch")
