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
