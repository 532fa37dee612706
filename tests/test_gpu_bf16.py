"""GPU parity of the bf16 MFMA contractions (C2/C4 precision).
Reference: fp64 convolution of the SAME bf16-rounded operands; the kernel accumulates in fp32 and rounds the result to
bf16 once, so the tolerance is one bf16 ulp of the output magnitude (2^-8 relative to max |y|) — stated per check."""
import pytest
import torch
import torch.nn.functional as F

from util import check

pytestmark = pytest.mark.gpu

BF16_TOL = 2.0 ** -8


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


CASES = [
    # N, Ci, H, W, Co, k, s, p
    (2, 64, 14, 14, 64, 3, 1, 1),
    (2, 64, 15, 13, 128, 3, 2, 1),
    (2, 128, 9, 9, 256, 1, 1, 0),
    (2, 256, 10, 10, 512, 1, 2, 0),
    (1, 96, 7, 7, 200, 3, 1, 1),
    (3, 32, 20, 18, 64, 1, 1, 0),
]


@pytest.mark.parametrize("case", CASES)
def test_conv_bf16_fwd_dgrad(edrl, dev, case):
    ops = edrl.ops
    N, Ci, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(21)
    x = torch.randn(N, H, W, Ci, generator=g).bfloat16()
    w = (torch.randn(Co, k, k, Ci, generator=g) * 0.1).bfloat16()
    xd = nchw(x.double()).requires_grad_(True)
    wd = w.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y = F.conv2d(xd, wd, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(dy.double())
    yh = ops.conv2d_fwd_bf16(x.to(dev), w.to(dev), s, p)
    assert yh.dtype == torch.bfloat16
    check(f"bf16 conv_fwd{case}", nchw(yh.float().cpu()), y, BF16_TOL)
    if Co % 32 == 0:
        wt = ops.permute_weight_bf16(w.float().to(dev))
        assert torch.equal(wt.cpu(), w.permute(3, 1, 2, 0).contiguous()), "bf16 weight permutation must be exact"
        dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
        dxh = ops.conv2d_dgrad_bf16(dyh, wt, (N, H, W, Ci), s, p)
        check(f"bf16 conv_dgrad{case}", nchw(dxh.float().cpu()), xd.grad, BF16_TOL)
        dx2 = ops.conv2d_dgrad_bf16(dyh, wt, (N, H, W, Ci), s, p, out=dxh.clone(), accumulate=True)
        check(f"bf16 conv_dgrad_accum{case}", nchw(dx2.float().cpu()), 2 * xd.grad, 2 * BF16_TOL)
    if Co % 8 == 0:
        dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
        dwh = ops.conv2d_wgrad_bf16(dyh, x.to(dev), (Co, k, k, Ci), s, p)      # fp32 result: only fp32 accumulation error
        check(f"bf16 conv_wgrad{case}", dwh.cpu().permute(0, 3, 1, 2), wd.grad, 2e-5)
        assert torch.equal(dwh, ops.conv2d_wgrad_bf16(dyh, x.to(dev), (Co, k, k, Ci), s, p)), "deterministic split-K"


def test_conv_bf16_exact_on_small_integers(edrl, dev):
    """Operand-layout check that cannot hide behind a tolerance: sparse 0/±1 data keeps every sum a small integer,
    exactly representable in bf16, so the result must be bit-exact."""
    ops = edrl.ops
    g = torch.Generator().manual_seed(22)
    N, Ci, H, W, Co = 2, 64, 12, 11, 128
    x = ((torch.rand(N, H, W, Ci, generator=g) < 0.05).float() * torch.randint(-1, 2, (N, H, W, Ci), generator=g)).bfloat16()
    w = ((torch.rand(Co, 3, 3, Ci, generator=g) < 0.2).float() * torch.randint(-1, 2, (Co, 3, 3, Ci), generator=g)).bfloat16()
    y = F.conv2d(nchw(x.double()), w.double().permute(0, 3, 1, 2), padding=1)
    assert y.abs().max() <= 128
    yh = ops.conv2d_fwd_bf16(x.to(dev), w.to(dev), 1, 1)
    assert torch.equal(nchw(yh.float().cpu()).double(), y), "bf16 MFMA operand/accumulator layout"


def test_conv_bf16_fused_stats_and_casts(edrl, dev):
    L = edrl._lib
    ops = edrl.ops
    g = torch.Generator().manual_seed(23)
    N, Ci, H, W, Co = 3, 64, 20, 18, 128
    x = (torch.randn(N, H, W, Ci, generator=g) + 0.5).to(dev)
    w = (torch.randn(Co, 3, 3, Ci, generator=g) * 0.1).to(dev)
    xb, wb = ops.to_bf16(x), ops.to_bf16(w)
    assert torch.equal(xb.cpu(), x.cpu().bfloat16()) and torch.equal(ops.to_f32(xb).cpu(), x.cpu().bfloat16().float())
    y, part, chunks = ops.conv2d_fwd_bf16(xb, wb, 1, 1, stats=True)
    yd = F.conv2d(nchw(xb.cpu().double()), wb.cpu().double().permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1).reshape(-1, Co)
    M = yd.shape[0]
    outs = [torch.empty(Co, device=dev) for _ in range(4)]
    gbytes = L.query("edrl_bn_finalize_group_ws_bytes", chunks, Co)
    gws = torch.empty(max(gbytes // 8, 1), device=dev, dtype=torch.float64)
    L.call("edrl_bn_finalize_partials_f32", L.ptr(part), chunks, 128, M, Co, None, None, None, None, 0.1, 1e-5,
           L.ptr(outs[0]), L.ptr(outs[1]), L.ptr(outs[2]), L.ptr(outs[3]), L.ptr(gws), gbytes)
    # statistics come from the fp32 accumulators (before the bf16 rounding of y): compare with the exact conv
    check("bf16 fused mean", outs[0].cpu(), yd.mean(0), 1e-4)
    check("bf16 fused rstd", outs[1].cpu(), 1.0 / torch.sqrt(yd.var(0, unbiased=False) + 1e-5), 1e-4)
