"""Per-conv check of the bf16 trunk building block against an fp64 storage-aware emulation with the SAME bf16 input."""
import sys, torch
import torch.nn.functional as F
sys.path.insert(0, ".")
import edrl_amd as edrl
from edrl_amd_pkg import encoders as E
dev = torch.device("cuda:0")
torch.manual_seed(0)
q = lambda t: t.to(torch.bfloat16).to(t.dtype)
def one(N, H, Ci, Co, k, stride, pad, relu, res):
    x = q(torch.randn(N, H, H, Ci).clamp_min(0))
    w = torch.randn(Co, k, k, Ci) * (2.0 / (k * k * Ci)) ** 0.5
    bn = dict(weight=torch.rand(Co) + 0.5, bias=torch.randn(Co) * 0.1, running_mean=torch.zeros(Co), running_var=torch.ones(Co), momentum=0.1, eps=1e-5)
    Ho = (H + 2 * pad - k) // stride + 1
    r = q(torch.randn(N, Ho, Ho, Co)) if res else None
    bnd = {kk: (v.to(dev) if torch.is_tensor(v) else v) for kk, v in bn.items()}
    raw, out, mean, rstd, mask = E._conv_bn_fwd_bf16(x.to(dev).bfloat16(), w.to(dev), bnd, stride, pad, relu, r.to(dev).bfloat16() if res else None)
    acc = F.conv2d(x.double().permute(0, 3, 1, 2), q(w).double().permute(0, 3, 1, 2), stride=stride, padding=pad)
    m = acc.mean((0, 2, 3), keepdim=True); v = acc.var((0, 2, 3), unbiased=False, keepdim=True)
    y = (q(acc) - m) * torch.rsqrt(v + 1e-5) * bn["weight"].double().view(1, -1, 1, 1) + bn["bias"].double().view(1, -1, 1, 1)
    if res: y = y + r.double().permute(0, 3, 1, 2)
    if relu: y = F.relu(y)
    y = q(y).permute(0, 2, 3, 1)
    o = out.float().cpu().double()
    e_raw = float((raw.float().cpu().double() - q(acc).permute(0, 2, 3, 1)).norm() / acc.norm())
    e_mean = float((mean.cpu().double() - m.flatten()).abs().max() / m.abs().max())
    e_rstd = float((rstd.cpu().double() * torch.sqrt(v.flatten() + 1e-5) - 1).abs().max())
    print(f"N{N} H{H} {Ci}->{Co} k{k} s{stride} relu{int(relu)} res{int(res)}: raw {e_raw:.2e} mean {e_mean:.2e} rstd {e_rstd:.2e} out fro {float((o - y).norm() / y.norm()):.2e}")
one(4, 24, 64, 64, 1, 1, 0, True, False)
one(4, 24, 64, 64, 3, 1, 1, True, False)
one(4, 24, 64, 256, 1, 1, 0, True, True)
one(4, 24, 256, 128, 1, 1, 0, True, False)
one(4, 24, 128, 128, 3, 2, 1, True, False)
one(4, 24, 256, 512, 1, 2, 0, False, False)
one(4, 6, 1024, 2048, 1, 2, 0, False, False)
one(4, 3, 512, 512, 3, 1, 1, True, False)
one(4, 3, 512, 2048, 1, 1, 0, True, True)
