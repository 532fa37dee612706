"""3-D-conv OCT encoder: volume operators and the ResNet3D trunk against torch-CPU fp64
(F.conv3d / F.batch_norm / F.max_pool3d).  Parity unpinned by the reference (its 3-D encoder source is absent)."""
import types

import pytest
import torch
import torch.nn.functional as F

from util import check

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", [
    # N, Ci, D, H, W, Co, k, stride, pad
    (2, 1, 9, 14, 12, 64, 7, 2, 3),      # stem: single channel, K padded 7 -> 8
    (2, 8, 6, 9, 9, 16, 3, 1, 1),
    (1, 16, 7, 10, 8, 32, 3, 2, 1),      # stride 2 in all three dims
    (2, 12, 5, 6, 6, 8, 1, 2, 0),        # 1x1x1 strided shortcut
    # Ci % 16 == 0: the forward decodes the depth taps inside the gather (edrl_conv3d_ndhwc_fwd_f32, no unfolded copy)
    (5, 16, 2, 4, 4, 16, 3, 1, 1),       # 32 rows per sample: a 128-row tile spans four samples (depth padding at every boundary)
    (2, 64, 4, 12, 12, 128, 3, 1, 1),
    (3, 32, 5, 6, 6, 64, 1, 2, 0),       # 1x1x1 stride-2 shortcut
    (2, 32, 9, 7, 5, 48, 3, 2, 1),       # stride 2 in all three dims, odd sizes
    (1, 128, 6, 14, 14, 256, 3, 1, 1),   # 128-wide N tiles
    # even depth: the data gradient decodes the depth taps in the gather as well (edrl_conv3d_ndhwc_dgrad_f32; depth parity classes)
    (2, 32, 8, 7, 5, 48, 3, 2, 1),       # stride 2 in all three dims, odd in-plane sizes
    (3, 32, 6, 6, 6, 64, 1, 2, 0),       # 1x1x1 stride-2 shortcut: the odd-depth class has no taps (zeros)
    (2, 64, 4, 8, 8, 128, 3, 2, 1),
    (2, 16, 2, 5, 5, 32, 3, 2, 1),       # one output depth
    (1, 64, 4, 6, 6, 144, 3, 1, 1),      # Co = 144: a K tail inside the last tap
])
def test_conv3d_fwd_dgrad_wgrad_vs_torch(edrl, dev, case):
    from edrl_amd_pkg.encoders3d import Conv3dFn
    N, Ci, D, H, W, Co, k, s, p = case
    torch.manual_seed(1)
    x = torch.randn(N, Ci, D, H, W)
    w = torch.randn(Co, Ci, k, k, k) * 0.1
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    y_ref = F.conv3d(xd, wd, stride=s, padding=p)
    gy = torch.randn(y_ref.shape)
    y_ref.backward(gy.double())
    ck = (k * Ci + 3) // 4 * 4
    wp = torch.zeros(Co, k, k, ck)
    wp[..., : k * Ci] = w.permute(0, 3, 4, 2, 1).reshape(Co, k, k, k * Ci)        # [Co,KH,KW,(KD,Ci)]
    xh = x.permute(0, 2, 3, 4, 1).contiguous().to(dev).requires_grad_(True)
    wh = wp.to(dev).requires_grad_(True)
    y = Conv3dFn.apply(xh, wh, k, s, s, p, p)
    check(f"conv3d_fwd{case}", y.detach().cpu().permute(0, 4, 1, 2, 3), y_ref.detach(), 2e-5)
    y.backward(gy.permute(0, 2, 3, 4, 1).contiguous().to(dev))
    check(f"conv3d_dgrad{case}", xh.grad.cpu().permute(0, 4, 1, 2, 3), xd.grad, 2e-5)
    dw = wh.grad.cpu()
    assert float(dw[..., k * Ci:].abs().max()) == 0.0 if ck > k * Ci else True      # padded K columns: exactly zero gradient
    check(f"conv3d_wgrad{case}", dw[..., : k * Ci].reshape(Co, k, k, k, Ci).permute(0, 4, 3, 1, 2), wd.grad, 2e-5)


def test_maxpool3d_bit_exact(edrl, dev):
    from edrl_amd_pkg.encoders3d import MaxPool3dFn
    torch.manual_seed(2)
    x = torch.randn(2, 8, 7, 11, 9)                     # N,C,D,H,W, odd sizes
    xd = x.double().requires_grad_(True)
    y_ref = F.max_pool3d(xd, 3, 2, 1)
    gy = torch.randn(y_ref.shape)
    y_ref.backward(gy.double())
    xh = x.permute(0, 2, 3, 4, 1).contiguous().to(dev).requires_grad_(True)
    y = MaxPool3dFn.apply(xh)
    assert torch.equal(y.detach().cpu().permute(0, 4, 1, 2, 3).double(), y_ref.detach())
    y.backward(gy.permute(0, 2, 3, 4, 1).contiguous().to(dev))
    check("maxpool3d_bwd", xh.grad.cpu().permute(0, 4, 1, 2, 3), xd.grad, 1e-6)


def test_resnet3d_trunk_fwd_bwd_vs_oracle(edrl, dev):
    """ResNet3D-10 trunk on [2,1,16,64,64]: feature map and all parameter gradients vs the fp64 oracle, within 1e-4 or
    5x the fp32 round-off envelope of the same computation (small-batch BatchNorm3d; same rule as the 2-D trunk tests)."""
    from oracle import resnet_oracle as RO
    from util import relerr
    torch.manual_seed(0)
    trunk = edrl.ResNet3DTrunk(10).to(dev).train()
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 1, 16, 64, 64, generator=g)
    layers = [1, 1, 1, 1]
    sd64 = RO.trunk3d_state(trunk)
    sd32 = RO.trunk3d_state(trunk, dtype=torch.float32)
    f64 = RO.trunk3d_forward(x.double(), sd64, layers)
    f32 = RO.trunk3d_forward(x, sd32, layers)
    gy = torch.randn(f64.shape, generator=g)
    f64.backward(gy.double())
    f32.backward(gy)
    f = trunk(x.view(2, 16, 64, 64, 1).to(dev))
    assert tuple(f.shape) == (2, 1, 2, 2, 512)
    env = relerr(f32, f64)
    check("trunk3d_fwd", f.detach().cpu().permute(0, 4, 1, 2, 3), f64.detach(), max(1e-4, min(5 * env, 1e-3)))
    f.backward(gy.permute(0, 2, 3, 4, 1).contiguous().to(dev))
    worst = 0.0
    for n, p in trunk.named_parameters():
        r = sd64[n].grad
        sc = r.abs().max().clamp_min(1e-12)
        e = float((p.grad.cpu().double() - r).abs().max() / sc)
        e32 = float((sd32[n].grad.double() - r).abs().max() / sc)
        worst = max(worst, e)
        assert e < max(5e-3, min(10 * e32, 2e-2)), f"grad {n}: rel err {e:.3e} (fp32 envelope {e32:.3e})"
    print(f"[parity] trunk3d: fwd envelope {env:.2e}, worst gradient rel err {worst:.3e}")
    assert int(trunk.bn1.num_batches_tracked) == 1
    check("trunk3d bn1.running_mean", trunk.bn1.running_mean.cpu(), sd64["bn1.running_mean"], 1e-5)
    # inference mode: running statistics (the state after the training pass above), nothing saved
    trunk.eval()
    with torch.no_grad():
        fe = trunk(x.view(2, 16, 64, 64, 1).to(dev))
        fe_ref = RO.trunk3d_forward(x.double(), RO.trunk3d_state(trunk, requires_grad=False), layers, train=False)
    check("trunk3d_eval_fwd", fe.cpu().permute(0, 4, 1, 2, 3), fe_ref, 1e-4)


def test_medfusion_step_with_3d_oct_encoder(edrl, dev):
    """args.oct_encoder="3d": the whole train_step runs through the 3-D-conv OCT encoder (tokens [B, d*h*w, 768])."""
    args = types.SimpleNamespace(mode="train", batch_size=2, encoder_depth=18, oct_encoder="3d", oct3d_depth=10)
    torch.manual_seed(0)
    m = edrl.MedFusion(2, 2, None, args).to(dev).train()
    opt = edrl.FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-6)
    data, y = edrl.synthetic_batch(2, 64, 64, 16, device=dev, seed=9)
    out = edrl.train_step(m, opt, data, y)
    assert torch.isfinite(out["loss"]) and out["pred"].shape == (2, 2)
    live = [n for n, p in m.named_parameters() if p.grad is not None and n.startswith("transformer_3DNet")]
    assert len(live) == len([n for n, _ in m.transformer_3DNet.named_parameters()])


@pytest.mark.parametrize("case", [
    # N, C, D, H, W, Co, k, stride, pad, relu, residual
    (2, 64, 4, 12, 12, 64, 3, 1, 1, True, True),
    (2, 64, 6, 10, 10, 128, 3, 2, 1, True, False),      # stride 2 in all three dims
    (3, 64, 4, 8, 8, 128, 1, 2, 0, False, False),       # 1x1x1 strided shortcut, no ReLU
    (1, 128, 3, 14, 14, 128, 3, 1, 1, True, True),
])
def test_conv_bn3d_bf16_unit_vs_fp64(edrl, dev, case):
    """The bf16 3-D unit (ConvBn3dBf16Fn: bf16 MFMA conv over the depth-unfolded bf16 operand -> BatchNorm3d(train) -> + residual ->
    ReLU) against fp64 on the SAME bf16-rounded inputs and weights: output within one bf16 rounding of the fp64 result plus the
    rounding of the stored raw tensor; gradients (input, weight, gamma, beta, residual) within the bf16 storage error of d_raw."""
    from edrl_amd_pkg.encoders3d import ConvBn3dBf16Fn
    N, C, D, H, W, Co, k, s, p, relu, has_res = case
    g = torch.Generator().manual_seed(4)
    xb = torch.randn(N, C, D, H, W, generator=g).bfloat16()
    w = (torch.randn(Co, C, k, k, k, generator=g) * 0.05)
    wb = w.bfloat16()
    gamma, beta = torch.rand(Co, generator=g) + 0.5, torch.randn(Co, generator=g) * 0.1
    xd = xb.double().requires_grad_(True)
    wd = wb.double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    raw = F.conv3d(xd, wd, stride=s, padding=p)
    rd = None
    if has_res:
        rb = torch.randn(raw.shape, generator=g).bfloat16()
        rd = rb.double().requires_grad_(True)
    # the kernel stores raw as bf16 and normalises the STORED values with the statistics of the fp32 accumulators: the same here
    # (rounding as a straight-through step), so that the ReLU decisions are taken on the same numbers up to accumulation order
    raw_q = raw + (raw.detach().bfloat16().double() - raw.detach())
    dims = (0, 2, 3, 4)
    mu = raw.mean(dims, keepdim=True)
    var = raw.var(dims, unbiased=False, keepdim=True)
    y = (raw_q - mu) / torch.sqrt(var + 1e-5) * gd.view(1, -1, 1, 1, 1) + bd.view(1, -1, 1, 1, 1)
    if has_res:
        y = y + rd
    if relu:
        y = F.relu(y)
    gy = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(gy.double())
    to5 = lambda t: t.permute(0, 2, 3, 4, 1).contiguous()
    xh = to5(xb).to(dev).requires_grad_(True)
    wh = wb.float().permute(0, 3, 4, 2, 1).reshape(Co, k, k, k * C).contiguous().to(dev).requires_grad_(True)     # [Co,KH,KW,(KD,Ci)], bf16-exact values
    gh, bh = gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
    rh = to5(rb).to(dev).requires_grad_(True) if has_res else None
    out = ConvBn3dBf16Fn.apply(xh, wh, gh, bh, rm, rv, k, s, s, p, p, relu, rh)
    assert out.dtype == torch.bfloat16
    check(f"bf16_unit3d_fwd{case}", out.detach().float().cpu().permute(0, 4, 1, 2, 3), y.detach(), 6e-3)
    out.backward(to5(gy).to(dev))

    def fro(name, got, ref, tol, tol_max):
        # relative Frobenius error + the worst entry relative to the largest reference entry (bf16 results: 2^-9 per rounding)
        e = float((got.double() - ref).norm() / ref.norm())
        em = float((got.double() - ref).abs().max() / ref.abs().max())
        print(f"[parity] {name}: rel Frobenius {e:.3e} (tol {tol:.0e}), worst entry {em:.3e} (tol {tol_max:.0e})")
        assert e < tol and em < tol_max, f"{name}: {e:.3e} / {em:.3e}"

    fro(f"bf16_unit3d_dx{case}", xh.grad.float().cpu().permute(0, 4, 1, 2, 3), xd.grad, 6e-3, 1.2e-2)
    fro(f"bf16_unit3d_dw{case}", wh.grad.cpu().reshape(Co, k, k, k, C).permute(0, 4, 3, 1, 2), wd.grad, 4e-3, 1e-2)
    fro(f"bf16_unit3d_dgamma{case}", gh.grad.cpu(), gd.grad, 1e-3, 2e-3)
    fro(f"bf16_unit3d_dbeta{case}", bh.grad.cpu(), bd.grad, 1e-3, 2e-3)
    if has_res:
        fro(f"bf16_unit3d_dres{case}", rh.grad.float().cpu().permute(0, 4, 1, 2, 3), rd.grad, 1e-3, 1e-1)     # (a flipped decision is a whole entry: none expected with the storage-aware reference)
    bs = raw.detach().transpose(0, 1).reshape(Co, -1)
    check("bf16_unit3d_running_mean", rm.cpu(), 0.1 * bs.mean(1), 2e-3)


def test_resnet3d_trunk_bf16_vs_fp32_trunk(edrl, dev):
    """ResNet3D-10 with bf16 residual stages against the fp32 trunk with the same parameters: features and parameter gradients
    agree at bf16 accuracy (relative to the tensor's largest element / by cosine), the pass is finite and bit-reproducible, and
    eval mode takes the fp32 path."""
    torch.manual_seed(0)
    t32 = edrl.ResNet3DTrunk(10).to(dev).train()
    t16 = edrl.ResNet3DTrunk(10, dtype="bf16").to(dev).train()
    t16.load_state_dict(t32.state_dict())
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 16, 64, 64, 1, generator=g).to(dev)
    gy = torch.randn(2, 1, 2, 2, 512, generator=g).to(dev)
    f32 = t32(x); f32.backward(gy)
    f16 = t16(x); f16.backward(gy)
    assert f16.dtype == torch.float32 and bool(torch.isfinite(f16).all())
    check("trunk3d_bf16_fwd", f16.detach().cpu(), f32.detach().cpu(), 6e-2)
    worst = 1.0
    for (n, p32), (_, p16) in zip(t32.named_parameters(), t16.named_parameters()):
        a, b = p16.grad.flatten().double(), p32.grad.flatten().double()
        assert bool(torch.isfinite(a).all()), n
        cos = float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))
        worst = min(worst, cos)
        assert cos > 0.93, f"{n}: cosine {cos:.4f}"      # (batch 2: small-batch BatchNorm amplifies the bf16 noise towards the stem)
    print(f"[parity] trunk3d bf16 vs fp32: worst gradient cosine {worst:.4f}")
    assert int(t16.blocks[0].bn1.num_batches_tracked) == 1
    t16b = edrl.ResNet3DTrunk(10, dtype="bf16").to(dev).train()
    t16b.load_state_dict(t32.state_dict())          # (t32's running statistics moved once; reset both to the same state)
    t16c = edrl.ResNet3DTrunk(10, dtype="bf16").to(dev).train()
    t16c.load_state_dict(t32.state_dict())
    fb = t16b(x); fb.backward(gy)
    fc = t16c(x); fc.backward(gy)
    assert torch.equal(fb, fc)
    for pb, pc in zip(t16b.parameters(), t16c.parameters()):
        assert torch.equal(pb.grad, pc.grad)
    t16.eval()
    t32.eval()
    with torch.no_grad():
        t16.load_state_dict(t32.state_dict())
        assert torch.equal(t16(x), t32(x))
