import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, edrl_amd
ops = edrl_amd.ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
N, H, Ci, Co = 8, 32, 256, 64          # dgrad of a 1x1 conv Ci -> Co: dx [N,H,H,Ci]
g = torch.randn(N, H, H, Co, device=dev); yraw = torch.randn(N, H, H, Co, device=dev)
bc = torch.randn(4, Co, device=dev) * 0.3
w = torch.randn(Co, 1, 1, Ci, device=dev) * 0.1
wt = ops.permute_weight(w)
x = torch.randn(N, H, H, Ci, device=dev)
fin = torch.randn(5, Ci, device=dev)
kb = torch.randint(0, 16, (N * H * H, Ci // 4), device=dev, dtype=torch.uint8)
old = torch.randn(N, H, H, Ci, device=dev)
d_raw = (bc[0] * g + bc[1] * yraw + bc[2]).double()
dx_ref = d_raw.view(-1, Co) @ w.view(Co, Ci).double() + old.view(-1, Ci).double()
bits = ((kb.view(-1, Ci // 4, 1).int() >> torch.arange(4, device=dev).view(1, 1, 4)) & 1).bool().view(-1, Ci)
dx_ref = dx_ref * bits
out = old.clone()
r, part, chunks = ops.conv2d_dgrad_bn(g, yraw, bc, wt, (N, H, H, Ci), 1, 0, out=out, accumulate=True, ep=(x, kb, fin, True))
got = r.view(-1, Ci).double()
err = (got - dx_ref).abs()
print("max err", float(err.max()), "ref max", float(dx_ref.abs().max()))
bad = err > 1e-3
print("bad", int(bad.sum()), "of", bad.numel(), "rows", bad.any(1).nonzero().flatten()[:10].tolist(), "cols", bad.any(0).nonzero().flatten()[:16].tolist())
r0 = int(bad.any(1).nonzero().flatten()[0]) if bad.any() else 0
print("row", r0, "got", got[r0, :8].tolist(), "\n ref", dx_ref[r0, :8].tolist(), "\n unmasked", (dx_ref / bits.clamp(min=1))[r0, :8].tolist(), "\n bits", bits[r0, :8].tolist(), "old", old.view(-1, Ci)[r0, :8].tolist())
s0 = part.view(chunks, 2, Ci)[:, 0].double().sum(0); s0r = dx_ref.sum(0)
s1 = part.view(chunks, 2, Ci)[:, 1].double().sum(0); s1r = (dx_ref * x.view(-1, Ci).double()).sum(0)
print("part err", float((s0 - s0r).abs().max()), float((s1 - s1r).abs().max()))
