import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, edrl_amd
dev = torch.device("cuda:0")
N, Ci, H, W, Co, k, s, p = 40, 16, 56, 56, 64, 1, 1, 0
g = torch.Generator().manual_seed(13)
x = (torch.randn(N, H, W, Ci, generator=g) + 0.7).to(dev)
w = (torch.randn(Co, k, k, Ci, generator=g) * 0.1).to(dev)
ref = (x.double().view(-1, Ci) @ w.double().view(Co, Ci).t()).float()
for it in range(3):
    y0 = edrl_amd.ops.conv2d_fwd(x, w, stride=s, pad=p).view(-1, Co)
    y1, part, chunks = edrl_amd.ops.conv2d_fwd_stats(x, w, None, s, p)
    y1 = y1.view(-1, Co)
    for nm, y in (("plain", y0), ("stats", y1)):
        bad = (y - ref).abs() > 1e-4
        rows = bad.any(1).nonzero().flatten()
        cols = bad.any(0).nonzero().flatten()
        print(it, nm, "bad elems", int(bad.sum()), "rows", rows[:12].tolist(), "n_rows", len(rows), "cols", cols[:16].tolist(), flush=True)
        if len(rows):
            r = int(rows[0]); print("   row", r, "got", y[r, :8].tolist(), "ref", ref[r, :8].tolist())
