"""Guard-band ("canary") test of every conv launcher on the small / ragged geometries of the ResNet-18 64x64 step (B=2, 4 OCT
slices: 8 and 2 images, feature maps 16x16 .. 2x2) -- the shapes at which a GPU memory access fault was once seen during round-1
development (gpurun_out/t5.log: test_full_train_step_vs_oracle, an uncommitted build of the v2 gather kernel; see DESIGN.md
"Fault record").  Every output tensor (conv result, BN partials, data gradient, weight gradient, split-K workspace) is carved
out of a larger buffer filled with a sentinel: after each launch the bytes in front of and behind it must be untouched, and
the result must match torch (fp64).  A launcher that writes outside its footprint at M < one tile, OH*OW < 16, Ktot == 0 parity
classes or ragged channel tails fails here deterministically instead of corrupting a neighbouring allocation."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
SENT = -7.25
PAD = 4096          # floats of guard band on each side (16 KiB)


class Guarded:
    def __init__(self, shape, dev, dtype=torch.float32):
        n = 1
        for s in shape:
            n *= s
        n4 = (n + 3) // 4 * 4
        self.big = torch.full((PAD + n4 + PAD,), SENT, device=dev, dtype=dtype)
        self.n = n
        self.t = self.big[PAD:PAD + n].view(shape)

    def check(self, what):
        head, tail = self.big[:PAD], self.big[PAD + self.n:]
        assert bool((head == SENT).all()) and bool((tail == SENT).all()), f"{what}: wrote outside its output tensor"


GEOMS = [  # N, H, W, Ci, Co, k, stride, pad
    (8, 16, 16, 64, 64, 3, 1, 1), (8, 16, 16, 64, 128, 3, 2, 1), (8, 16, 16, 64, 128, 1, 2, 0), (8, 8, 8, 128, 128, 3, 1, 1),
    (8, 8, 8, 128, 256, 3, 2, 1), (8, 8, 8, 128, 256, 1, 2, 0), (8, 4, 4, 256, 512, 3, 2, 1), (8, 2, 2, 512, 512, 3, 1, 1),
    (2, 16, 16, 64, 64, 3, 1, 1), (2, 4, 4, 256, 512, 3, 2, 1), (2, 2, 2, 512, 512, 3, 1, 1), (2, 2, 2, 256, 512, 1, 2, 0),
    (3, 7, 5, 64, 256, 1, 1, 0), (3, 7, 5, 256, 64, 1, 1, 0), (1, 13, 11, 64, 64, 3, 2, 1), (5, 9, 9, 16, 20, 3, 1, 1),
    # full 128-row tiles (the lean buffer-store epilogue) with a channel count that ends inside a tile: the column guard is an
    # out-of-range lane offset there, the row advance a scalar offset
    (8, 16, 16, 64, 208, 1, 1, 0), (8, 16, 16, 208, 64, 3, 1, 1), (8, 16, 16, 80, 208, 3, 2, 1),
]


@pytest.mark.parametrize("N,H,W,Ci,Co,k,s,p", GEOMS)
def test_conv_launchers_stay_inside_their_outputs(edrl, dev, N, H, W, Ci, Co, k, s, p):
    ops, L = edrl.ops, edrl._lib
    P = L.ptr
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Co)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(N, H, W, Ci, generator=g)
    w = torch.randn(Co, k, k, Ci, generator=g) * 0.1
    dy = torch.randn(N, Ho, Wo, Co, generator=g)
    xd, wd, dyd = x.to(dev), w.to(dev), dy.to(dev)
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    y64 = F.conv2d(x64, w64, stride=s, padding=p)
    y64.backward(dy.double().permute(0, 3, 1, 2))
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    # forward (plain epilogue)
    y = Guarded((N, Ho, Wo, Co), dev)
    ops.conv2d_fwd(xd, wd, stride=s, pad=p, out=y.t)
    torch.cuda.synchronize(); y.check("conv fwd")
    assert rel(y.t.permute(0, 3, 1, 2), y64.detach()) < 2e-5
    # forward with the BatchNorm partials from the epilogue
    if Ci % 16 == 0 and Co % 4 == 0:
        chunks = L.query("edrl_conv_stats_chunks", N, Ho, Wo)
        y2, part = Guarded((N, Ho, Wo, Co), dev), Guarded((chunks, 3, Co), dev)
        L.call("edrl_conv2d_nhwc_fwd_stats_f32", P(xd), P(wd), P(y2.t), None, P(part.t), part.t.numel() * 4, N, H, W, Ci, Ho, Wo, Co,
               k, k, s, p)
        torch.cuda.synchronize(); y2.check("conv fwd+stats (y)"); part.check("conv fwd+stats (partials)")
        assert torch.equal(y2.t, y.t)
    # data gradient (parity classes for stride 2), write and accumulate
    if Ci % 4 == 0:
        wt = ops.permute_weight(wd)
        dx = Guarded((N, H, W, Ci), dev)
        ops.conv2d_dgrad(dyd, wt, (N, H, W, Ci), s, p, out=dx.t)
        torch.cuda.synchronize(); dx.check("conv dgrad")
        assert rel(dx.t.permute(0, 3, 1, 2), x64.grad) < 2e-5
        ops.conv2d_dgrad(dyd, wt, (N, H, W, Ci), s, p, out=dx.t, accumulate=True)
        torch.cuda.synchronize(); dx.check("conv dgrad (accumulate)")
        assert rel(dx.t.permute(0, 3, 1, 2), 2 * x64.grad) < 2e-5
    # weight gradient incl. its split-K workspace
    nbytes = L.query("edrl_conv2d_nhwc_wgrad_workspace_bytes", N, Ho, Wo, Co, Ci, k, k)
    dw, ws = Guarded((Co, k, k, Ci), dev), Guarded((nbytes // 4,), dev)
    L.call("edrl_conv2d_nhwc_wgrad_f32", P(dyd), P(xd), P(dw.t), P(ws.t), nbytes, N, H, W, Ci, Ho, Wo, Co, k, k, s, p, Co, Ci, 0)
    torch.cuda.synchronize(); dw.check("conv wgrad"); ws.check("conv wgrad (split-K workspace)")
    assert rel(dw.t.permute(0, 3, 1, 2), w64.grad) < 2e-5
    # fused-BatchNorm variants where the geometry has them
    if ops.conv_fused_ok(N, H, W, Ci, Co, k, s, p):
        fin = torch.empty(5, Ci, device=dev); fin[0].normal_(generator=None); fin[1].fill_(1.0); fin[2].uniform_(0.5, 1.5)
        fin[3].normal_(); fin[4] = fin[3] - fin[0] * fin[2]
        chunks = L.query("edrl_conv_stats_chunks", N, Ho, Wo)
        y3, part3 = Guarded((N, Ho, Wo, Co), dev), Guarded((chunks, 3, Co), dev)
        L.call("edrl_conv2d_nhwc_fwd_bnin_stats_f32", P(xd), P(fin), P(wd), P(y3.t), P(part3.t), part3.t.numel() * 4, N, H, W, Ci,
               Ho, Wo, Co, k, k, s, p)
        torch.cuda.synchronize(); y3.check("fused fwd (y)"); part3.check("fused fwd (partials)")
        act = torch.relu(x.double() * fin[2].cpu().double() + fin[4].cpu().double()).permute(0, 3, 1, 2)
        assert rel(y3.t.permute(0, 3, 1, 2), F.conv2d(act, w64.detach(), stride=s, padding=p)) < 2e-5
        bc = torch.empty(4, Co, device=dev); bc[0].uniform_(0.5, 1.5); bc[1].normal_().mul_(0.01); bc[2].normal_().mul_(0.01); bc[3].zero_()
        yraw = torch.randn(N, Ho, Wo, Co, device=dev)
        nch = L.query("edrl_conv_dgrad_bn_chunks", N, H, W, s, p)
        dx2, part4 = Guarded((N, H, W, Ci), dev), Guarded((nch, 2, Ci), dev)
        wt = ops.permute_weight(wd)
        L.call("edrl_conv2d_nhwc_dgrad_bn_f32", P(dyd), P(yraw), P(bc), P(wt), P(dx2.t), N, H, W, Ci, Ho, Wo, Co, k, k, s, p, 0,
               P(xd), None, P(fin), 1, P(part4.t), part4.t.numel() * 4)
        torch.cuda.synchronize(); dx2.check("fused dgrad (dx)"); part4.check("fused dgrad (partials)")
        dw2, ws2 = Guarded((Co, k, k, Ci), dev), Guarded((nbytes // 4,), dev)
        L.call("edrl_conv2d_nhwc_wgrad_bn_f32", P(dyd), P(yraw), P(bc), P(xd), P(fin), P(dw2.t), P(ws2.t), nbytes, N, H, W, Ci, Ho, Wo,
               Co, k, k, s, p, 0)
        torch.cuda.synchronize(); dw2.check("fused wgrad"); ws2.check("fused wgrad (workspace)")
        draw = (bc[0] * dyd + bc[1] * yraw + bc[2]).double().cpu().permute(0, 3, 1, 2)
        a64 = act.clone().requires_grad_(True)
        w65 = w64.detach().clone().requires_grad_(True)
        F.conv2d(a64, w65, stride=s, padding=p).backward(draw)
        assert rel(dw2.t.permute(0, 3, 1, 2), w65.grad) < 5e-5
        keep = (x.double() * fin[2].cpu().double() + fin[4].cpu().double() > 0).permute(0, 3, 1, 2)
        pre = (x.double() * fin[2].cpu().double() + fin[4].cpu().double()).permute(0, 3, 1, 2)
        care = pre.abs() > 1e-6 * pre.abs().max()
        err = ((dx2.t.double().cpu().permute(0, 3, 1, 2) - a64.grad * keep) * care).abs().max() / a64.grad.abs().max()
        assert err < 5e-5, err


BF16_GEOMS = [  # N, H, W, Ci, Co, k, stride, pad : channel counts the 256x256 LDS-DMA cores accept (multiples of 256) at row counts
    # from 12 (far below one tile) to 686 (two and a bit), ragged last tiles, padding taps, stride-2 parity classes (incl. one that
    # no tap reaches), pixel counts that are not multiples of the weight gradient's 32-pixel unit
    (2, 7, 5, 256, 256, 3, 1, 1), (3, 4, 4, 256, 512, 1, 2, 0), (1, 9, 7, 512, 256, 3, 2, 1), (3, 2, 2, 256, 256, 3, 1, 1),
    (14, 7, 7, 256, 256, 1, 1, 0), (2, 6, 6, 512, 512, 1, 2, 0),
    # the 128-row / 128x128 kernels' domain: narrow layers, channel counts that end inside a tile
    (2, 9, 9, 64, 64, 3, 1, 1), (3, 7, 5, 64, 200, 1, 1, 0), (2, 8, 8, 96, 64, 3, 2, 1),
    # the weight-stationary kernels' domain (conv_c64_bf16.hip): 64 -> 64 3x3 (above), 64 -> 256 and 128 -> 512 1x1; pixel counts off
    # their 16 / 64 / 128-pixel granularities
    (3, 7, 5, 64, 256, 1, 1, 0), (2, 6, 6, 128, 512, 1, 1, 0), (1, 11, 13, 64, 64, 3, 1, 1),
]


@pytest.mark.parametrize("mode", ["0", "2", "v3s"])
@pytest.mark.parametrize("N,H,W,Ci,Co,k,s,p", BF16_GEOMS + [(2, 7, 5, 128, 128, 3, 1, 1), (3, 5, 3, 128, 384, 1, 2, 0), (5, 9, 9, 128, 128, 1, 1, 0)])
def test_bf16_conv_launchers_stay_inside_their_outputs(edrl, dev, N, H, W, Ci, Co, k, s, p, mode, switches):
    """The same guard-band check for the bf16 launchers: forward (+ BatchNorm partials), data gradient (write / accumulate), weight
    gradient incl. its split-K workspace -- once on the 128-row / 128x128 kernels (mode 0) and once with the 256x256 LDS-DMA cores and
    the weight-stationary kernels forced wherever the geometry allows (mode 2: conv_bf16_v3.hip, conv_wgrad_bf16_v3.hip,
    conv_c64_bf16.hip; their loads for rows / taps outside the tensors are out-of-range buffer offsets or clamped rows and must
    neither fault nor leak into the outputs)."""
    ops, L = edrl.ops, edrl._lib
    P = L.ptr
    if mode == "v3s":          # the 128x128 small-tile LDS-DMA core (conv_bf16_v3s.hip) forced wherever its geometry allows
        switches(EDRL_BF16_V3S="2", EDRL_BF16_V3="0", EDRL_BF16_WGRAD_V3="0", EDRL_BF16_C64="0", EDRL_BF16_K64="0")
    else:
        switches(EDRL_BF16_V3S="0")
        switches(EDRL_BF16_V3=mode)
        switches(EDRL_BF16_WGRAD_V3=mode)
        switches(EDRL_BF16_C64=mode)
        switches(EDRL_BF16_K64=mode)
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Co)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    x = torch.randn(N, H, W, Ci, generator=g).to(bf)
    w = (torch.randn(Co, k, k, Ci, generator=g) * 0.1).to(bf)
    dy = torch.randn(N, Ho, Wo, Co, generator=g).to(bf)
    xd, wd, dyd = x.to(dev), w.to(dev), dy.to(dev)
    x64 = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    w64 = w.double().permute(0, 3, 1, 2).requires_grad_(True)
    y64 = F.conv2d(x64, w64, stride=s, padding=p)
    y64.backward(dy.double().permute(0, 3, 1, 2))
    rel = lambda a, b: ((a.double().cpu() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
    tol = 2.0 ** -8
    if Ci % 32 == 0 and Co % 4 == 0:
        chunks = L.query("edrl_conv_stats_chunks", N, Ho, Wo)
        y, part = Guarded((N, Ho, Wo, Co), dev, bf), Guarded((chunks, 3, Co), dev)
        L.call("edrl_conv2d_nhwc_fwd_bf16", P(xd), P(wd), P(y.t), P(part.t), part.t.numel() * 4, N, H, W, Ci, Ho, Wo, Co, k, k, s, p)
        torch.cuda.synchronize(); y.check("bf16 conv fwd (y)"); part.check("bf16 conv fwd (partials)")
        assert rel(y.t.permute(0, 3, 1, 2), y64.detach()) < tol
    if Co % 32 == 0 and Ci % 4 == 0:
        wt = ops.permute_weight_bf16(wd.float())
        dx = Guarded((N, H, W, Ci), dev, bf)
        dx.t.zero_()          # (a stride-2 class that no tap reaches is left as it is)
        L.call("edrl_conv2d_nhwc_dgrad_bf16", P(dyd), P(wt), P(dx.t), N, H, W, Ci, Ho, Wo, Co, k, k, s, p, 0)
        torch.cuda.synchronize(); dx.check("bf16 conv dgrad")
        assert rel(dx.t.permute(0, 3, 1, 2), x64.grad) < tol
        L.call("edrl_conv2d_nhwc_dgrad_bf16", P(dyd), P(wt), P(dx.t), N, H, W, Ci, Ho, Wo, Co, k, k, s, p, 2)
        torch.cuda.synchronize(); dx.check("bf16 conv dgrad (accumulate)")
        assert rel(dx.t.permute(0, 3, 1, 2), 2 * x64.grad) < 2 * tol
    if Co % 8 == 0 and Ci % 8 == 0:
        nbytes = L.query("edrl_conv2d_nhwc_wgrad_bf16_workspace_bytes", N, Ho, Wo, Co, Ci, k, k)
        dw, ws = Guarded((Co, k, k, Ci), dev), Guarded((nbytes // 4,), dev)
        L.call("edrl_conv2d_nhwc_wgrad_bf16", P(dyd), P(xd), P(dw.t), P(ws.t), nbytes, N, H, W, Ci, Ho, Wo, Co, k, k, s, p, 0)
        torch.cuda.synchronize(); dw.check("bf16 conv wgrad"); ws.check("bf16 conv wgrad (split-K workspace)")
        assert rel(dw.t.permute(0, 3, 1, 2), w64.grad) < 2e-5
