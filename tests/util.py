import torch
import torch.nn.functional as F


def relerr(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def check(name, got, ref, tol):
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    e = relerr(got, ref)
    print(f"[parity] {name}: max-rel-err {e:.3e} (tol {tol:.1e})")
    assert e == e and e <= tol, f"{name}: rel err {e:.3e} > {tol:.1e}"


def _nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def large_mean_case(N, H, W, Ci, Co, k, s, p, seed, bf16=False):
    """Operands of one fused conv <- BatchNorm(+ReLU) backward unit whose BatchNorm input has |mean| / sigma = 50 in every channel
    (x = 0.1 randn +- 5): raw tensor x of the BatchNorm below, the conv above it and its output gradient, the fp64 reference
    gradients (dgamma, dbeta, d_x) through relu(bn(x)) -> conv, and fcoef [5][Ci] from the fp64 batch statistics."""
    g = torch.Generator().manual_seed(seed)
    sign = torch.where(torch.rand(Ci, generator=g) < 0.5, -1.0, 1.0)
    x = 0.1 * torch.randn(N, H, W, Ci, generator=g) + 5.0 * sign
    w = torch.randn(Co, k, k, Ci, generator=g) * 0.1
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    dy = torch.randn(N, Ho, Wo, Co, generator=g)
    gamma = 0.5 + torch.rand(Ci, generator=g)
    beta = 0.3 * torch.randn(Ci, generator=g)
    if bf16:
        x, w, dy = x.bfloat16().float(), w.bfloat16().float(), dy.bfloat16().float()
    xd = _nchw(x.double()).requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    pre = F.batch_norm(xd, None, None, gd, bd, training=True, eps=1e-5)
    out = F.conv2d(torch.relu(pre), _nchw(w.double()), stride=s, padding=p)
    out.backward(_nchw(dy.double()))
    x2 = x.double().reshape(-1, Ci)
    mean, var = x2.mean(0), x2.var(0, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    scale = gamma.double() * rstd
    fc = torch.stack([mean, rstd, scale, beta.double(), beta.double() - mean * scale]).float()
    care = (pre.detach().abs() > 1e-3 * pre.detach().abs().max())      # decisions within round-off of zero may differ
    return x, w, dy, gamma, fc, (gd.grad, bd.grad, xd.grad), pre.detach(), care
