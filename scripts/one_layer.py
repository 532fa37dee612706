"""Run one conv layer shape repeatedly (for rocprofv3 --pmc passes). usage: one_layer.py Ci H Co k s p [images] [mode]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, edrl_amd
ops = edrl_amd.ops
Ci, H, Co, k, s, p = map(int, sys.argv[1:7])
N = int(sys.argv[7]) if len(sys.argv) > 7 else 256
mode = sys.argv[8] if len(sys.argv) > 8 else "fwd"
dev = torch.device("cuda:0")
Ho = (H + 2 * p - k) // s + 1
x = torch.randn(N, H, H, Ci, device=dev); w = torch.randn(Co, k, k, Ci, device=dev) * 0.05
dy = torch.randn(N, Ho, Ho, Co, device=dev); wt = ops.permute_weight(w)
y = torch.empty(N, Ho, Ho, Co, device=dev); dx = torch.empty_like(x); dw = torch.empty_like(w)
for _ in range(10):
    if mode == "fwd": ops.conv2d_fwd(x, w, stride=s, pad=p, out=y)
    elif mode == "dgrad": ops.conv2d_dgrad(dy, wt, tuple(x.shape), s, p, out=dx)
    else: ops.conv2d_wgrad(dy, x, tuple(w.shape), s, p, out=dw)
torch.cuda.synchronize()
print("done", 2.0 * N * Ho * Ho * Co * k * k * Ci / 1e9, "GFLOP per launch")
