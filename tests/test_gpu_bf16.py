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


V3_CASES = [
    # N, Ci, H, W, Co, k, s, p : Co (forward) and Ci (data gradient) multiples of 256 where the v3 core must take the call;
    # ragged row counts (N*Ho*Wo not a multiple of 256, one case below a single tile), padding taps, stride-2 parity classes,
    # K from 64 (two ring units, shorter than the 3-deep prefetch) to 2304
    (3, 256, 14, 14, 256, 3, 1, 1),
    (2, 256, 13, 11, 512, 3, 2, 1),
    (5, 64, 9, 7, 256, 1, 1, 0),
    (2, 512, 10, 10, 256, 1, 2, 0),
    (1, 256, 7, 7, 256, 3, 1, 1),
    (4, 1024, 6, 5, 256, 1, 1, 0),
    (16, 64, 70, 70, 256, 1, 1, 0),      # 307 tiles: more than one tile per workgroup in the persistent form (forward)
    (20, 256, 64, 64, 256, 1, 1, 0),     # 320 tiles, forward + data gradient (plain and accumulating)
    (9, 256, 62, 62, 256, 3, 2, 1),      # stride 2: four parity classes of 34 tiles each, ragged
]


@pytest.mark.parametrize("persist", ["1", "0"])
@pytest.mark.parametrize("case", V3_CASES)
def test_conv_bf16_v3_core_vs_fp64_and_v2(edrl, dev, case, persist, switches):
    """The 256x256 LDS-DMA core (csrc/conv_bf16_v3.hip), forced on (EDRL_BF16_V3=2) at sizes far below its production range:
    forward (+ fused BatchNorm chunk partials), data gradient (plain, accumulating, stride-2 parity classes) against the fp64
    convolution of the same bf16 operands at one bf16 ulp of the output range, the chunk partials against sums taken from the
    kernel's own output rows in fp64 (1e-3 of the per-chunk scale: they come from the unrounded fp32 accumulators), and against
    the 128-row kernel (EDRL_BF16_V3=0), which must agree to one bf16 ulp (both accumulate k in the same order in fp32; on the
    production shapes they are bit-identical, scripts/v3_layer_bench.py)."""
    ops = edrl.ops
    N, Ci, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(33)
    x = torch.randn(N, H, W, Ci, generator=g).bfloat16()
    w = (torch.randn(Co, k, k, Ci, generator=g) * 0.1).bfloat16()
    xd = nchw(x.double()).requires_grad_(True)
    wd = w.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y = F.conv2d(xd, wd, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(dy.double())
    Ho, Wo = y.shape[2], y.shape[3]
    outs = {}
    switches(EDRL_BF16_V3_PERSIST=persist)      # "1": the persistent form (conv_bf16_v3p.hip: register epilogue, next tile prefetched)
    switches(EDRL_BF16_V3S="0")                 # (the small-tile core has its own test below)
    for mode in ("0", "2"):
        switches(EDRL_BF16_V3=mode)
        yh, part, chunks = ops.conv2d_fwd_bf16(x.to(dev), w.to(dev), s, p, stats=True)
        dxh = dxa = None
        if Ci % 256 == 0:
            wt = ops.permute_weight_bf16(w.float().to(dev))
            dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
            dxh = ops.conv2d_dgrad_bf16(dyh, wt, (N, H, W, Ci), s, p)
            dxa = ops.conv2d_dgrad_bf16(dyh, wt, (N, H, W, Ci), s, p, out=dxh.clone(), accumulate=True)
        torch.cuda.synchronize()
        outs[mode] = (yh.float().cpu(), part.cpu(), dxh.float().cpu() if dxh is not None else None,
                      dxa.float().cpu() if dxa is not None else None)
    yv, pv, dv, dav = outs["2"]
    check(f"v3 conv_fwd{case}", nchw(yv), y, BF16_TOL)
    check(f"v3 vs v2 conv_fwd{case}", yv, outs["0"][0], BF16_TOL)
    # chunk partials [chunks][3][Co]: S1 = sum (y - K), S2 = sum (y - K)^2 over the chunk's rows, K = its first row
    rows = y.detach().permute(0, 2, 3, 1).reshape(-1, Co)
    M = rows.shape[0]
    assert pv.shape[0] == (M + 127) // 128
    for c in range(pv.shape[0]):
        blk = rows[c * 128:min(M, (c + 1) * 128)]
        K = pv[c, 2].double()
        assert float((K - blk[0]).abs().max()) <= BF16_TOL * float(rows.abs().max()), "shift = the chunk's first row (fp32 accumulator)"
        d = blk - K
        sc = max(float(d.abs().sum(0).max()), 1e-6)
        assert float((pv[c, 0].double() - d.sum(0)).abs().max()) <= 1e-3 * sc, f"chunk {c} S1"
        sc2 = max(float((d * d).sum(0).max()), 1e-6)
        assert float((pv[c, 1].double() - (d * d).sum(0)).abs().max()) <= 1e-3 * sc2, f"chunk {c} S2"
    if dv is not None:
        check(f"v3 conv_dgrad{case}", nchw(dv), xd.grad, BF16_TOL)
        check(f"v3 conv_dgrad_accum{case}", nchw(dav), 2 * xd.grad, 2 * BF16_TOL)
        check(f"v3 vs v2 conv_dgrad{case}", dv, outs["0"][2], BF16_TOL)


@pytest.mark.parametrize("N,H,W", [(2, 14, 14), (3, 9, 7), (1, 2, 2), (5, 16, 13), (1, 23, 3)])
def test_conv3x3_c64_kernel_vs_fp64_and_128row_kernel(edrl, dev, N, H, W, switches):
    """The weight-stationary 64 -> 64 3x3 kernel (csrc/conv_c64_bf16.hip; weights in LDS in fragment order, pixel fragments straight
    from global memory, lane-pair swap for 16-byte stores), forced on (EDRL_BF16_C64=2) at sizes far below its production range --
    pixel counts that are not multiples of 16 / 64 / 128, 2x2 images (every tap but the centre row / column masked), several images
    per block -- forward with BatchNorm chunk partials, plain data gradient, and the data gradient with the masked-gradient epilogue,
    against fp64 (one bf16 ulp of the output range; partial sums 1e-3 of their scale: they come from the unrounded accumulators)
    and against the 128-row kernel (EDRL_BF16_C64=0)."""
    ops, L = edrl.ops, edrl._lib
    P = L.ptr
    C = 64
    g = torch.Generator().manual_seed(41)
    x = torch.randn(N, H, W, C, generator=g).bfloat16()
    w = (torch.randn(C, 3, 3, C, generator=g) * 0.1).bfloat16()
    xd = nchw(x.double()).requires_grad_(True)
    wd = w.double().permute(0, 3, 1, 2).contiguous()
    y = F.conv2d(xd, wd, padding=1)
    dy = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(dy.double())
    M = N * H * W
    xraw = torch.randn(N, H, W, C, generator=g).bfloat16()                    # raw tensor of the BatchNorm below (epilogue operand)
    mask = torch.randint(0, 16, (M, C // 4), generator=g, dtype=torch.uint8)
    fc = torch.ones(5, C)
    outs = {}
    for mode in ("0", "2"):
        switches(EDRL_BF16_C64=mode)
        yh, part, chunks = ops.conv2d_fwd_bf16(x.to(dev), w.to(dev), 1, 1, stats=True)
        wt = ops.permute_weight_bf16(w.float().to(dev))
        dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
        dxh = ops.conv2d_dgrad_bf16(dyh, wt, (N, H, W, C), 1, 1)
        gm, epart, _ = ops.conv2d_dgrad_bn_bf16(dyh, None, None, wt, (N, H, W, C), 1, 1, ep=(xraw.to(dev), mask.to(dev), fc.to(dev), True))
        torch.cuda.synchronize()
        outs[mode] = (yh.float().cpu(), part.cpu(), dxh.float().cpu(), gm.float().cpu(), epart.cpu())
    yv, pv, dv, gv, ev = outs["2"]
    check(f"c64 fwd {N}x{H}x{W}", nchw(yv), y, BF16_TOL)
    check(f"c64 fwd vs 128-row {N}x{H}x{W}", yv, outs["0"][0], BF16_TOL)
    rows = y.detach().permute(0, 2, 3, 1).reshape(-1, C)
    assert pv.shape[0] == (M + 127) // 128
    for c in range(pv.shape[0]):
        blk = rows[c * 128:min(M, (c + 1) * 128)]
        K = pv[c, 2].double()
        assert float((K - blk[0]).abs().max()) <= BF16_TOL * float(rows.abs().max()), "shift = the chunk's first row"
        d = blk - K
        assert float((pv[c, 0].double() - d.sum(0)).abs().max()) <= 1e-3 * max(float(d.abs().sum(0).max()), 1e-6), f"chunk {c} S1"
        assert float((pv[c, 1].double() - (d * d).sum(0)).abs().max()) <= 1e-3 * max(float((d * d).sum(0).max()), 1e-6), f"chunk {c} S2"
    check(f"c64 dgrad {N}x{H}x{W}", nchw(dv), xd.grad, BF16_TOL)
    check(f"c64 dgrad vs 128-row {N}x{H}x{W}", dv, outs["0"][2], BF16_TOL)
    keep = ((mask.view(M, C // 4, 1).int() >> torch.arange(4).view(1, 1, 4)) & 1).view(N, H, W, C).double()
    gref = xd.grad.permute(0, 2, 3, 1) * keep
    check(f"c64 masked dgrad {N}x{H}x{W}", gv.double(), gref, BF16_TOL)
    for c in range(ev.shape[0]):
        gb = gref.reshape(-1, C)[c * 128:min(M, (c + 1) * 128)]
        xb = xraw.double().reshape(-1, C)[c * 128:min(M, (c + 1) * 128)]
        assert float((ev[c, 0].double() - gb.sum(0)).abs().max()) <= 1e-3 * max(float(gb.abs().sum(0).max()), 1e-6), f"chunk {c} sum g"
        xs = xb - fc[0].double()          # plane 1 is taken shifted by the batch mean (fcoef row 0)
        assert float((ev[c, 1].double() - (gb * xs).sum(0)).abs().max()) <= 1e-3 * max(float((gb * xs).abs().sum(0).max()), 1e-6), f"chunk {c} sum g*(x-mean)"


@pytest.mark.parametrize("M_hw,C,Co", [((2, 14, 14), 64, 256), ((3, 9, 7), 64, 128), ((1, 5, 5), 64, 64), ((5, 16, 13), 64, 512),
                                       ((2, 14, 14), 128, 512), ((3, 9, 7), 128, 128), ((5, 16, 13), 128, 256)])
def test_conv1x1_k64_streaming_kernel_vs_fp64_and_128row_kernel(edrl, dev, M_hw, C, Co, switches):
    """The streaming 64 | 128 -> Co 1x1 kernel (csrc/conv_c64_bf16.hip: weights in LDS, a wave owns 128 pixels x all output channels, pixel
    fragments loaded and transformed in registers once per chunk), forced on (EDRL_BF16_K64=2) at small sizes (pixel counts off every
    tile size, one case below a single chunk): plain forward with BatchNorm chunk partials and the fused form (BatchNorm + ReLU of the
    input in the operand), against fp64 of the same bf16 operands (one bf16 ulp; partials 1e-3 of their scale) and against the
    128-row kernel (EDRL_BF16_K64=0), which forms the same transformed operand (same fp32 arithmetic, one rounding)."""
    ops, L = edrl.ops, edrl._lib
    N, H, W = M_hw
    g = torch.Generator().manual_seed(43)
    x = torch.randn(N, H, W, C, generator=g).bfloat16()
    w = (torch.randn(Co, 1, 1, C, generator=g) * 0.2).bfloat16()
    fc = torch.zeros(5, C)
    fc[2] = 0.5 + torch.rand(C, generator=g); fc[4] = 0.3 * torch.randn(C, generator=g)
    outs = {}
    for mode in ("0", "2"):
        switches(EDRL_BF16_K64=mode)
        y0, p0, ch = ops.conv2d_fwd_bf16(x.to(dev), w.to(dev), 1, 0, stats=True)
        y1, p1, _ = ops.conv2d_fwd_bnin_stats_bf16(x.to(dev), fc.to(dev), w.to(dev), 1, 0)
        torch.cuda.synchronize()
        outs[mode] = (y0.float().cpu(), p0.cpu(), y1.float().cpu(), p1.cpu())
    M = N * H * W
    wd = w.double().view(Co, C)
    act = torch.relu(x.float() * fc[2] + fc[4]).bfloat16().double().view(M, C)      # the operand the kernels form: fp32 fma, one rounding
    for name, xin, (yv, pv), (yo, _) in (("plain", x.double().view(M, C), outs["2"][0:2], outs["0"][0:2]),
                                         ("fused", act, outs["2"][2:4], outs["0"][2:4])):
        ref = xin @ wd.t()
        check(f"k64 {name} {M_hw}->{Co}", yv.view(M, Co), ref, BF16_TOL)
        check(f"k64 {name} vs 128-row {M_hw}->{Co}", yv, yo, BF16_TOL)
        assert pv.shape[0] == (M + 127) // 128
        for c in range(pv.shape[0]):
            blk = ref[c * 128:min(M, (c + 1) * 128)]
            K = pv[c, 2].double()
            assert float((K - blk[0]).abs().max()) <= BF16_TOL * float(ref.abs().max()), "shift = the chunk's first row"
            d = blk - K
            assert float((pv[c, 0].double() - d.sum(0)).abs().max()) <= 1e-3 * max(float(d.abs().sum(0).max()), 1e-6), f"{name} chunk {c} S1"
            assert float((pv[c, 1].double() - (d * d).sum(0)).abs().max()) <= 1e-3 * max(float((d * d).sum(0).max()), 1e-6), f"{name} chunk {c} S2"


W3_CASES = [
    # N, Ci, H, W, Co, k, s, p : both channel counts multiples of 256 (the v3 weight-gradient core's domain); pixel counts that
    # are not multiples of the 32-pixel unit, fewer units than the 3-deep prefetch, padding taps, stride 2 (odd sizes: the last
    # input row / column is never read), several tiles along both output axes, images shorter than one unit (4 x 3 = 12 pixels)
    (3, 256, 14, 14, 256, 3, 1, 1),
    (2, 256, 13, 11, 512, 3, 2, 1),
    (5, 256, 9, 7, 256, 1, 1, 0),
    (2, 512, 10, 10, 256, 1, 2, 0),
    (1, 256, 7, 7, 256, 3, 1, 1),
    (7, 1024, 4, 3, 256, 1, 1, 0),
    (40, 256, 7, 7, 512, 3, 1, 1),
]


@pytest.mark.parametrize("case", W3_CASES)
def test_conv_wgrad_bf16_v3_core_vs_fp64_and_v2(edrl, dev, case, switches):
    """The 256x256 LDS-DMA weight-gradient core (csrc/conv_wgrad_bf16_v3.hip: transposing LDS reads of DMA-written pixel-major
    images, X-row offsets decoded once per workgroup into an LDS table), forced on (EDRL_BF16_WGRAD_V3=2) far below its production
    range, against the fp64 weight gradient of the same bf16 operands (fp32 result: only fp32 accumulation error, 2e-5 of the
    output's max, as the 128x128 kernel), against that kernel (EDRL_BF16_WGRAD_V3=0; different split plan, so not bit-identical),
    accumulating into an existing gradient, and run-to-run deterministic."""
    ops = edrl.ops
    N, Ci, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(35)
    x = torch.randn(N, H, W, Ci, generator=g).bfloat16()
    w = (torch.randn(Co, k, k, Ci, generator=g) * 0.1).bfloat16()
    xd = nchw(x.double())
    wd = w.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y = F.conv2d(xd, wd, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(dy.double())
    dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
    xh = x.to(dev)
    outs = {}
    for mode in ("0", "2"):
        switches(EDRL_BF16_WGRAD_V3=mode)
        dw = ops.conv2d_wgrad_bf16(dyh, xh, (Co, k, k, Ci), s, p)
        dw_again = ops.conv2d_wgrad_bf16(dyh, xh, (Co, k, k, Ci), s, p)
        dwa = ops.conv2d_wgrad_bf16(dyh, xh, (Co, k, k, Ci), s, p, out=dw.clone(), accumulate=True)
        torch.cuda.synchronize()
        assert torch.equal(dw, dw_again), "deterministic split-K"
        outs[mode] = (dw.cpu(), dwa.cpu())
    dv, dav = outs["2"]
    check(f"v3 conv_wgrad{case}", dv.permute(0, 3, 1, 2), wd.grad, 2e-5)
    check(f"v3 conv_wgrad_accum{case}", dav.permute(0, 3, 1, 2), 2 * wd.grad, 4e-5)
    check(f"v3 vs v2 conv_wgrad{case}", dv, outs["0"][0], 4e-5)


@pytest.mark.parametrize("v8", ["0", "1"])
def test_fused_stem_mx_matches_fp32_kernels(edrl, dev, v8, switches):
    """The bf16 trunk's stem (BatchNorm + ReLU folded into the 3x3/s2 max-pool, edrl_maxpool3x3s2_bn_*_mx): the pooled tensor is
    the fp32 kernel's result rounded to bf16 ONCE (bit-exact), the arg-max bytes are identical, and the two backward kernels fed
    a bf16 gradient equal the fp32 kernels fed the same values as fp32, bit for bit (same arithmetic behind an exact widening
    load).  Odd sizes: ragged pooling windows at the right / bottom edge, a row count that is not a multiple of the chunk."""
    L = edrl._lib
    P = L.ptr
    # v8 = 1: the all-bf16 calls below run the 8-channels-per-thread kernels (EDRL_STEM_POOL_V8, the default): forward and d_raw
    # stay bit-identical; their partial sums are taken in another order (32 row lanes instead of 16), hence 1e-6 instead of equality
    switches(EDRL_STEM_POOL_V8=v8)
    N, H, W, C = 3, 37, 29, 64
    g = torch.Generator().manual_seed(8)
    raw = torch.randn(N, H, W, C, generator=g).to(dev)
    fc = torch.empty(5, C, device=dev)
    fc[0] = 0.1 * torch.randn(C, generator=g).to(dev); fc[1] = 1.0 + 0.1 * torch.rand(C, generator=g).to(dev)
    fc[2] = fc[1] * (0.5 + torch.rand(C, generator=g).to(dev)); fc[3] = 0.1 * torch.randn(C, generator=g).to(dev)
    fc[4] = fc[3] - fc[0] * fc[2]
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    y32 = torch.empty(N, Ho, Wo, C, device=dev); i32 = torch.empty(N, Ho, Wo, C, device=dev, dtype=torch.uint8)
    y16 = torch.empty(N, Ho, Wo, C, device=dev, dtype=torch.bfloat16); i16 = torch.empty_like(i32)
    L.call("edrl_maxpool3x3s2_bn_fwd_f32", P(raw), P(fc), P(y32), P(i32), N, H, W, C)
    L.call("edrl_maxpool3x3s2_bn_fwd_mx", P(raw), 0, P(fc), P(y16), 1, P(i16), N, H, W, C)
    assert torch.equal(y16, y32.bfloat16()) and torch.equal(i16, i32)
    dy16 = torch.randn(N, Ho, Wo, C, generator=g).bfloat16().to(dev)
    dy32 = dy16.float()
    M = N * H * W
    nb = L.query("edrl_bn_workspace_bytes", M, C)
    ws32 = torch.zeros(nb // 4, device=dev); ws16 = torch.zeros(nb // 4, device=dev)
    L.call("edrl_maxpool3x3s2_bn_bwd_reduce_f32", P(dy32), P(i32), P(raw), P(fc), P(ws32), nb, N, H, W, C)
    L.call("edrl_maxpool3x3s2_bn_bwd_reduce_mx", P(dy16), 1, P(i32), P(raw), 0, P(fc), P(ws16), nb, N, H, W, C)
    chunks = (M + 1023) // 1024
    assert torch.equal(ws16[:chunks * 3 * C].view(chunks, 3, C)[:, :2], ws32[:chunks * 3 * C].view(chunks, 3, C)[:, :2])
    bc = torch.empty(4, C, device=dev)
    bc[0] = 0.5 + torch.rand(C, generator=g).to(dev); bc[1] = 0.01 * torch.randn(C, generator=g).to(dev)
    bc[2] = 0.01 * torch.randn(C, generator=g).to(dev); bc[3] = fc[0]
    d32 = torch.empty_like(raw); d16 = torch.empty_like(raw)
    L.call("edrl_maxpool3x3s2_bn_bwd_apply_f32", P(dy32), P(i32), P(raw), P(fc), P(bc), P(d32), N, H, W, C)
    L.call("edrl_maxpool3x3s2_bn_bwd_apply_mx", P(dy16), 1, P(i32), P(raw), 0, P(fc), P(bc), P(d16), 0, N, H, W, C)
    assert torch.equal(d16, d32)
    # raw stem output stored as bf16 (x_bf16 = 1, the bf16 trunk's default): the kernels fed the bf16 tensor equal the fp32
    # kernels fed the same values widened to fp32, bit for bit
    raw16 = raw.bfloat16(); raw16f = raw16.float()
    y16b = torch.empty_like(y16); i16b = torch.empty_like(i32)
    L.call("edrl_maxpool3x3s2_bn_fwd_f32", P(raw16f), P(fc), P(y32), P(i32), N, H, W, C)
    L.call("edrl_maxpool3x3s2_bn_fwd_mx", P(raw16), 1, P(fc), P(y16b), 1, P(i16b), N, H, W, C)
    assert torch.equal(y16b, y32.bfloat16()) and torch.equal(i16b, i32)
    ws32.zero_(); ws16.zero_()
    L.call("edrl_maxpool3x3s2_bn_bwd_reduce_f32", P(dy32), P(i32), P(raw16f), P(fc), P(ws32), nb, N, H, W, C)
    L.call("edrl_maxpool3x3s2_bn_bwd_reduce_mx", P(dy16), 1, P(i32), P(raw16), 1, P(fc), P(ws16), nb, N, H, W, C)
    pa, pb = ws16[:chunks * 3 * C].view(chunks, 3, C)[:, :2], ws32[:chunks * 3 * C].view(chunks, 3, C)[:, :2]
    if v8 == "0":
        assert torch.equal(pa, pb)
    else:
        assert float((pa - pb).abs().max()) <= 1e-6 * float(pb.abs().max())
    L.call("edrl_maxpool3x3s2_bn_bwd_apply_f32", P(dy32), P(i32), P(raw16f), P(fc), P(bc), P(d32), N, H, W, C)
    L.call("edrl_maxpool3x3s2_bn_bwd_apply_mx", P(dy16), 1, P(i32), P(raw16), 1, P(fc), P(bc), P(d16), 0, N, H, W, C)
    assert torch.equal(d16, d32)
    dh = torch.empty(N, H, W, C, device=dev, dtype=torch.bfloat16)          # d_raw stored as bf16: the same values rounded once
    L.call("edrl_maxpool3x3s2_bn_bwd_apply_mx", P(dy16), 1, P(i32), P(raw16), 1, P(fc), P(bc), P(dh), 1, N, H, W, C)
    assert torch.equal(dh, d32.bfloat16())
    assert L.lib().fn["edrl_maxpool3x3s2_bn_fwd_mx"](P(raw16), 1, P(fc), P(y32), 0, P(i16b), N, H, W, C, L.stream()) == -22, \
        "a bf16 raw tensor needs a bf16 pooled tensor"
    L.call("edrl_maxpool3x3s2_bn_fwd_f32", P(raw), P(fc), P(y32), P(i32), N, H, W, C)      # (restored for the torch check below)
    # and the fp32 pair itself against torch: max-pool of relu(x*scale + shift2), gradient routed to the first arg-max
    act = torch.relu(torch.addcmul(fc[4], raw, fc[2])).permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    ref = F.max_pool2d(act, 3, 2, 1)
    check("fused stem fwd vs torch", y32.permute(0, 3, 1, 2).cpu(), ref, 1e-6)


@pytest.mark.parametrize("in_ch,H,W", [(1, 64, 48), (3, 32, 64), (1, 33, 47)])
def test_stem_conv_obf16_matches_fp32_kernel(edrl, dev, in_ch, H, W):
    """The bf16 trunk's stem conv (edrl_conv2d_nhwc_fwd_stats_f32_obf16: fp32 image, fp32 MFMA, bf16 store, BatchNorm partials from
    the accumulators): its output is the fp32 stem conv's output rounded to bf16 ONCE, bit for bit (same tiles, same accumulation
    order), and the finalised statistics are those of the UNROUNDED fp64 conv (1e-4 / 1e-4 relative) -- space-to-depth geometry
    (even sizes; 1 and 3 image channels) and the plain 7x7/s2 fallback (odd sizes), ragged last tile."""
    ops = edrl.ops
    L = edrl._lib
    N, Co = 3, 64
    g = torch.Generator().manual_seed(17)
    cp = 4
    x = torch.zeros(N, H, W, cp)
    x[..., :in_ch] = torch.rand(N, H, W, in_ch, generator=g)
    w = torch.zeros(Co, 7, 7, cp)
    w[..., :in_ch] = 0.1 * torch.randn(Co, 7, 7, in_ch, generator=g)
    xh, wh = x.to(dev), w.to(dev)
    y32, _, _ = ops.stem_conv_fwd(xh, wh)
    y16, part, chunks, _, folded = ops.stem_conv_fwd_obf16(xh, wh)
    assert folded == (H % 2 == 0 and W % 2 == 0)
    assert y16.dtype == torch.bfloat16 and torch.equal(y16, y32.bfloat16())
    M = y16.numel() // Co
    outs = [torch.empty(Co, device=dev) for _ in range(4)]
    gbytes = L.query("edrl_bn_finalize_group_ws_bytes", chunks, Co)
    gws = torch.empty(max(gbytes // 8, 1), device=dev, dtype=torch.float64)
    L.call("edrl_bn_finalize_partials_f32", L.ptr(part), chunks, 128, M, Co, None, None, None, None, 0.1, 1e-5,
           L.ptr(outs[0]), L.ptr(outs[1]), L.ptr(outs[2]), L.ptr(outs[3]), L.ptr(gws), gbytes)
    yd = F.conv2d(nchw(x.double()), w.double().permute(0, 3, 1, 2), stride=2, padding=3).permute(0, 2, 3, 1).reshape(-1, Co)
    check("stem obf16 mean", outs[0].cpu(), yd.mean(0), 1e-4)
    check("stem obf16 rstd", outs[1].cpu(), 1.0 / torch.sqrt(yd.var(0, unbiased=False) + 1e-5), 1e-4)
    # weight gradient with a bf16 d_raw (edrl_conv2d_nhwc_wgrad_f32_dybf16) == the fp32 kernel fed the widened values, bit for bit
    _, xk, _ = ops.stem_conv_fwd(xh, wh)
    dy16 = torch.randn(y16.shape, generator=g).bfloat16().to(dev)
    dw16 = ops.stem_conv_wgrad(dy16, xk, tuple(w.shape), folded)
    dw32 = ops.stem_conv_wgrad(dy16.float(), xk, tuple(w.shape), folded)
    assert torch.equal(dw16, dw32)


def test_bn_draw_bf16_kernel(edrl, dev):
    """edrl_bn_draw_bf16: d_raw = A*g + nK2*x + C2 per channel on bf16 tensors (fp32 arithmetic, one rounding), the standalone
    form of what the fused consumers build in their operand loads -- power-of-two width (chunked path) and an odd one (generic)."""
    L = edrl._lib
    P = L.ptr
    for M, C in ((3001, 128), (517, 24)):
        g = torch.Generator().manual_seed(9)
        gg = torch.randn(M, C, generator=g).bfloat16().to(dev); x = torch.randn(M, C, generator=g).bfloat16().to(dev)
        bc = torch.randn(4, C, generator=g).to(dev)
        out = torch.empty_like(gg)
        L.call("edrl_bn_draw_bf16", P(gg), P(x), P(bc), P(out), M, C)
        ref = (bc[0].double() * gg.double() + bc[1].double() * x.double() + bc[2].double())
        check(f"bn_draw_bf16 {M}x{C}", out.float().cpu(), ref.cpu(), BF16_TOL)


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


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("depth,in_ch,N,H,dropped", [(18, 1, 8, 96, False), (50, 3, 4, 128, False),
                                                     (18, 1, 8, 96, True), (50, 1, 4, 128, True)])
def test_trunk_bf16_per_layer_and_envelope(edrl, dev, depth, in_ch, N, H, dropped, fused, monkeypatch):
    """bf16 trunk (bf16 MFMA convs, bf16 activations/gradients, fp32 BN statistics) against the fp64 oracle.

    dropped=True is config C4's missing-modality view (BASELINE.json configs[4]; data_harvard.py:333-334: the OCT volume of
    the second view is all zeros): the stem conv output is identically 0, so the first BatchNorm sees EXACTLY zero variance
    (rstd = eps^-1/2, output = relu(beta)) and every later one sees only the variance the zero padding of a constant image
    creates -- through edrl_bn_apply_mx / edrl_bn_bwd_mx.  Random non-zero beta/gamma keep the pass non-trivial.

    fused=False runs the separate BatchNorm passes (EDRL_FUSE_BN=0) so that the per-layer hook (1) sees every conv+BN call;
    fused=True runs the default fused-BatchNorm blocks (BN+ReLU in the consumer conv's operand load, BatchNorm backward from the
    data-gradient epilogue, d_raw formed in the dgrad / wgrad loads) and is held to the same end-to-end criteria (2) and (3).

    (1) PER LAYER, tight: every conv->BN->(+res)->(ReLU) call the trunk makes is re-done by the oracle's storage-aware
        fp64 op (oracle/resnet_oracle.conv_bn_bf16_op) on the product's own bf16 inputs.  What is left is accumulation
        order plus 1-ulp(bf16) rounding flips of a ~1e-4 fraction of the elements: conv output and block output within
        3e-4 norm-wise (Frobenius), batch mean within 1e-5 of its max, rstd within 1e-5 relative.
    (2) END TO END, envelope: a rounding turns a perturbation d into an error ~sqrt(d*ulp), so two bf16-storage
        pipelines decorrelate with depth up to the bf16-storage drift itself; the product's distance to the UNROUNDED
        fp64 trunk must stay within 1.25x the distance the storage-aware oracle itself shows (+1e-3).
    (3) Parameter gradients: finite, and for every tensor whose direction survives bf16 storage at all (the
        storage-aware oracle's straight-through gradient has cosine > 0.9 to the fp64 trunk's), the product's cosine
        is no worse than the oracle's minus 0.08."""
    from oracle import resnet_oracle as RO
    import edrl_amd_pkg.encoders as E
    torch.manual_seed(0)
    trunk = edrl.ResNetTrunk(depth, in_ch, dtype="bf16").to(dev).train()
    last_bn = ".bn3.weight" if trunk.kind == "bottleneck" else ".bn2.weight"
    with torch.no_grad():                       # damp the residual branches (as zero-init-residual training recipes do):
        for n, p in trunk.named_parameters():   # a gamma=1 random-init stack is chaotic at these tiny batch sizes
            if n.endswith(last_bn):
                p.fill_(0.25)
        if dropped:
            gb = torch.Generator().manual_seed(7)
            for n, p in trunk.named_parameters():
                if n.endswith(".bias"):
                    p.copy_((0.3 * torch.randn(p.shape, generator=gb) + 0.2).to(dev))
    g = torch.Generator().manual_seed(1)
    x = torch.rand(N, in_ch, H, H, generator=g)
    if dropped:
        x.zero_()
    sd_full = RO.trunk_state(trunk)
    sd_q = RO.trunk_state(trunk)
    f_full = RO.trunk_forward(x.double(), sd_full, trunk.kind, trunk.blocks)
    f_q = RO.trunk_forward_bf16(x.double(), sd_q, trunk.kind, trunk.blocks)
    gy = torch.randn(f_full.shape, generator=g)
    f_full.backward(gy.double())
    f_q.backward(gy.double())

    monkeypatch.setattr(E, "_FUSE_BN", fused)
    product_op = E._conv_bn_fwd_bf16
    worst = dict(raw=0.0, out=0.0, mean=0.0, rstd=0.0, n=0)

    def checked(inp, w, bn, stride, pad, relu, residual=None):
        res = product_op(inp, w, bn, stride, pad, relu, residual)
        raw, out, mean, rstd, _ = res
        nchw = lambda t: t.float().cpu().double().permute(0, 3, 1, 2)
        r_raw, r_out, r_mean, r_var = RO.conv_bn_bf16_op(
            nchw(inp), w.detach().cpu().double(), bn["weight"].detach().cpu().double(),
            bn["bias"].detach().cpu().double(), stride, pad, relu, None if residual is None else nchw(residual))
        worst["raw"] = max(worst["raw"], float((nchw(raw) - r_raw).norm() / r_raw.norm().clamp_min(1e-30)))
        worst["out"] = max(worst["out"], float((nchw(out) - r_out).norm() / r_out.norm().clamp_min(1e-30)))
        worst["mean"] = max(worst["mean"], float((mean.cpu().double() - r_mean).abs().max() / r_mean.abs().max().clamp_min(1e-3)))
        worst["rstd"] = max(worst["rstd"], float((rstd.cpu().double() * torch.sqrt(r_var + 1e-5) - 1).abs().max()))
        worst["n"] += 1
        return res

    monkeypatch.setattr(E, "_conv_bn_fwd_bf16", checked)
    cp = trunk.in_ch_padded
    xh = torch.zeros(N, H, H, cp)
    xh[..., :in_ch] = x.permute(0, 2, 3, 1)
    f = trunk(xh.to(dev))
    monkeypatch.setattr(E, "_conv_bn_fwd_bf16", product_op)
    assert f.dtype == torch.float32
    n_convs = sum(1 for n in trunk.param_names if n.endswith(".weight") and "conv" in n or "downsample.0" in n) - 1
    print(f"[parity] bf16 trunk{depth}: {worst['n']} conv+BN layers checked per layer: raw {worst['raw']:.2e} "
          f"out {worst['out']:.2e} (tol 3e-4)  mean {worst['mean']:.2e} rstd {worst['rstd']:.2e} (tol 1e-5)")
    if fused:
        assert worst["n"] < n_convs, "the fused blocks must not go through the separate conv+BN path"
    else:
        assert worst["n"] == n_convs
    # dropped view: a layer's input is (nearly) one value per channel, so a 1-ulp(bf16) rounding flip of that value moves
    # thousands of equal elements together -- the norm-wise bound is looser there (1e-3; one bf16 ulp is 3.9e-3)
    tol = 1e-3 if dropped else 3e-4
    if worst["n"]:
        assert worst["raw"] < tol and worst["out"] < tol and worst["mean"] < 1e-5 and worst["rstd"] < 1e-5

    fh = f.permute(0, 3, 1, 2).cpu().double()
    rel = lambda a, b: float((a - b).norm() / b.norm())
    e_prod, e_env = rel(fh, f_full.detach()), rel(f_q.detach(), f_full.detach())
    print(f"[parity] bf16 trunk{depth}_fwd: distance to fp64 trunk {e_prod:.3e}; storage-aware oracle's own {e_env:.3e}")
    assert e_prod < 1.25 * e_env + 1e-3

    f.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    cosf = lambda a, b: float((a @ b) / (a.norm() * b.norm()).clamp_min(1e-30))
    margin, wn, n_checked = 1.0, "", 0
    for n, p in trunk.named_parameters():
        ref = sd_full[n].grad.flatten()
        got = p.grad.cpu().double().flatten()
        assert torch.isfinite(got).all(), n
        if ref.norm() < 1e-6 * max(1.0, float(sd_full[n].detach().norm())):
            continue
        if cosf(sd_q[n].grad.flatten(), ref) < 0.9:
            continue
        n_checked += 1
        m = cosf(got, ref) - cosf(sd_q[n].grad.flatten(), ref)
        if m < margin:
            margin, wn = m, f"{n}: product {cosf(got, ref):.4f} oracle {cosf(sd_q[n].grad.flatten(), ref):.4f}"
    print(f"[parity] bf16 trunk{depth} gradients: {n_checked} tensors, worst cosine margin vs storage-aware oracle "
          f"{margin:+.4f} ({wn})")
    # dropped view: with (near-)zero-variance BatchNorm (rstd ~ eps^-1/2 amplification) hardly any gradient DIRECTION survives bf16
    # storage even in the storage-aware fp64 oracle (its own cosine to the fp64 trunk is < 0.9 for most tensors), so only
    # finiteness and the margin on the surviving tensors bind there
    # margin: two bf16-storage pipelines with different rounding points (the fused path stores the masked gradient once, rounds
    # d_raw in the consumer's operand load and reduces from the unrounded fp32 values) differ from each other by the bf16 noise
    # itself on the most upstream tensors (bn1.bias after 53 layers at N=4: product 0.87-0.89, oracle 0.93); the kernels
    # themselves are pinned tightly by test_bf16_fused_bn_conv_kernels_vs_fp64 / the per-layer check above
    assert margin > -0.08 and n_checked > (0 if dropped else len(trunk.param_names) // 2)


def test_bn_mx_kernels_vs_torch(edrl, dev):
    """edrl_bn_apply_mx / edrl_bn_bwd_mx (bf16 and fp32 raw storage) against fp64 torch autograd on the same bf16
    inputs: outputs are bf16, so element-wise tolerance 2^-8 of the tensor max; dgamma/dbeta (fp32 reductions) 1e-4."""
    from edrl_amd_pkg import _lib as L
    from edrl_amd_pkg import encoders as E
    P = L.ptr
    torch.manual_seed(3)
    M, C = 4 * 14 * 14, 256
    for raw_dtype in (torch.bfloat16, torch.float32):
        raw = (torch.randn(M, C) * 1.5 + 0.3).to(raw_dtype)
        res = torch.randn(M, C).bfloat16()
        gamma = torch.rand(C) + 0.5
        beta = torch.randn(C) * 0.1
        mean = raw.double().mean(0)
        var = raw.double().var(0, unbiased=False)
        rstd = torch.rsqrt(var + 1e-5)
        scale = (gamma.double() * rstd).float()
        shift = beta.clone()
        d = lambda t: None if t is None else t.to(dev)
        out = torch.empty(M, C, device=dev, dtype=torch.bfloat16)
        mask = torch.empty(M, C // 4, device=dev, dtype=torch.uint8)
        rawd, resd = d(raw), d(res)
        meand, rstdd, scaled, shiftd = d(mean.float()), d(rstd.float()), d(scale), d(shift)
        L.call("edrl_bn_apply_mx", P(rawd), int(raw_dtype == torch.bfloat16), P(meand), P(scaled), P(shiftd),
               P(resd), P(out), 1, P(mask), M, C, 1)
        rawr = raw.double().requires_grad_(True)
        resr = res.double().requires_grad_(True)
        y = torch.relu((rawr - mean) * rstd * gamma.double() + beta.double() + resr)
        check(f"bn_apply_mx[{raw_dtype}]", out.float().cpu(), y.detach(), 2 ** -8)
        dy = torch.randn(M, C).bfloat16()
        gd = gamma.double().requires_grad_(True)
        bd = beta.double().requires_grad_(True)
        m2 = rawr.mean(0)
        v2 = rawr.var(0, unbiased=False)
        y2 = torch.relu((rawr - m2) * torch.rsqrt(v2 + 1e-5) * gd + bd + resr)
        y2.backward(dy.double())
        dyd, gammad = d(dy), d(gamma)
        d_raw, dg, db, dres = E._bn_bwd_mx(dyd, mask, rawd, meand, rstdd, gammad, True)
        assert d_raw.dtype == raw_dtype
        check(f"bn_bwd_mx[{raw_dtype}] d_raw", d_raw.float().cpu(), rawr.grad, 2 ** -8)
        check(f"bn_bwd_mx[{raw_dtype}] dres", dres.float().cpu(), resr.grad, 2 ** -8)
        check(f"bn_bwd_mx[{raw_dtype}] dgamma", dg.cpu(), gd.grad, 1e-4)
        check(f"bn_bwd_mx[{raw_dtype}] dbeta", db.cpu(), bd.grad, 1e-4)
    # bf16 max-pool: exact on bf16 values
    x = torch.randn(2, 17, 19, 64).bfloat16()
    N_, H_, W_, C_ = x.shape
    Ho, Wo = (H_ - 1) // 2 + 1, (W_ - 1) // 2 + 1
    xd = x.to(dev)
    o = torch.empty(N_, Ho, Wo, C_, device=dev, dtype=torch.bfloat16)
    idx = torch.empty(N_, Ho, Wo, C_, device=dev, dtype=torch.uint8)
    L.call("edrl_maxpool3x3s2_fwd_bf16", P(xd), P(o), P(idx), N_, H_, W_, C_)
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    yr = torch.nn.functional.max_pool2d(xr, 3, 2, 1)
    assert torch.equal(o.float().cpu().double().permute(0, 3, 1, 2), yr.detach())
    go = torch.randn(N_, Ho, Wo, C_).bfloat16()
    yr.backward(go.double().permute(0, 3, 1, 2))
    dxd = torch.empty_like(xd)
    god = go.to(dev)
    L.call("edrl_maxpool3x3s2_bwd_bf16", P(god), P(idx), P(dxd), N_, H_, W_, C_)
    check("maxpool_bwd_bf16", dxd.float().cpu().permute(0, 3, 1, 2), xr.grad, 2 ** -8)


@pytest.mark.parametrize("N,H,W,Ci,Co,k,s,p", [(4, 14, 14, 64, 128, 3, 1, 1), (3, 12, 10, 128, 64, 1, 1, 0), (2, 16, 16, 64, 128, 3, 2, 1),
                                               (2, 8, 8, 256, 512, 1, 2, 0)])
def test_bf16_fused_bn_conv_kernels_vs_fp64(edrl, dev, N, H, W, Ci, Co, k, s, p):
    """The fused-BatchNorm bf16 launchers against fp64 on the same bf16 operands with the product's storage roundings: forward with
    BN+ReLU in the operand load (+ BatchNorm partials of the output), data gradient with d_raw formed in the operand load and the
    masked result + (sum g, sum g*x) from the epilogue, weight gradient with both operand transforms.  bf16 outputs: 2^-7 of the
    tensor max element-wise (one rounding of the operand + one of the result); fp32 sums / weight gradients 2e-3."""
    ops, L = edrl.ops, edrl._lib
    P = L.ptr
    if not ops.conv_fused_ok_bf16(N, H, W, Ci, Co, k, s, p):
        pytest.skip("geometry without the fused bf16 fast paths")
    g = torch.Generator().manual_seed(N * 100 + Co)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    q = lambda t: t.bfloat16()
    x = q(torch.randn(N, H, W, Ci, generator=g))                 # raw conv output of the layer below (bf16 storage)
    w = torch.randn(Co, k, k, Ci, generator=g) * 0.1
    fin = torch.empty(5, Ci); fin[0].normal_(generator=g); fin[1].fill_(1.0); fin[2] = 0.5 + torch.rand(Ci, generator=g)
    fin[3].normal_(generator=g); fin[4] = fin[3] - fin[0] * fin[2]
    nchw = lambda t: t.double().permute(0, 3, 1, 2)
    act = q(torch.relu(x.float() * fin[2] + fin[4]))             # what the consumer forms on the fly, rounded to bf16 storage
    wq = q(w)
    y64 = F.conv2d(nchw(act), wq.double().permute(0, 3, 1, 2), stride=s, padding=p)
    xd, find = x.to(dev), fin.to(dev)
    y, part, chunks = ops.conv2d_fwd_bnin_stats_bf16(xd, find, wq.to(dev), s, p)
    check("bf16 fused fwd", nchw(y.float().cpu()), y64, 2 ** -7)
    # BatchNorm partials of the (unrounded) output -> mean / rstd
    fc = torch.empty(5, Co, device=dev)
    gb = L.query("edrl_bn_finalize_group_ws_bytes", chunks, Co)
    gws = torch.empty(max(gb // 8, 1), device=dev, dtype=torch.float64)
    L.call("edrl_bn_finalize_fcoef_f32", P(part), chunks, 128, N * Ho * Wo, Co, None, None, None, None, 0.1, 1e-5, P(fc), P(gws), gb)
    check("bf16 fused fwd mean", fc[0].cpu(), y64.mean(dim=(0, 2, 3)), 1e-3)
    check("bf16 fused fwd rstd", fc[1].cpu(), torch.rsqrt(y64.var(dim=(0, 2, 3), unbiased=False) + 1e-5), 1e-3)
    # ---- backward operands
    gy = q(torch.randn(N, Ho, Wo, Co, generator=g))              # masked upstream gradient g of this layer's BatchNorm output
    yraw = q(torch.randn(N, Ho, Wo, Co, generator=g))            # this layer's raw conv output
    bc = torch.empty(4, Co); bc[0] = 0.5 + torch.rand(Co, generator=g); bc[1] = 0.02 * torch.randn(Co, generator=g)
    bc[2] = 0.02 * torch.randn(Co, generator=g); bc[3].zero_()
    draw = q(bc[0] * gy.float() + bc[1] * yraw.float() + bc[2])  # d_raw as the consumers form it, bf16 storage
    a64 = nchw(act).requires_grad_(True)
    w64 = wq.double().permute(0, 3, 1, 2).requires_grad_(True)
    F.conv2d(a64, w64, stride=s, padding=p).backward(nchw(draw))
    gyd, yrawd, bcd = gy.to(dev), yraw.to(dev), bc.to(dev)
    wt = ops.permute_weight_bf16(w.to(dev))
    dx, part2, ch2 = ops.conv2d_dgrad_bn_bf16(gyd, yrawd, bcd, wt, (N, H, W, Ci), s, p, ep=(xd, None, find, True))
    pre = nchw(x.float() * fin[2] + fin[4])
    keep = (pre > 0).double()
    care = (pre.abs() > 1e-6 * pre.abs().max()).double()
    ref_dx = a64.grad * keep
    err = ((nchw(dx.float().cpu()) - ref_dx) * care).abs().max() / ref_dx.abs().max()
    print(f"[parity] bf16 fused dgrad+epilogue: max-rel-err {err:.3e} (tol {2 ** -7:.1e})")
    assert err <= 2 ** -7
    xs = nchw(x) - fin[0].double().view(1, -1, 1, 1)          # plane 1 of the partials: sum g*(x - mean)
    s_g = (ref_dx * care).sum(dim=(0, 2, 3)); s_gx = (ref_dx * care * xs).sum(dim=(0, 2, 3))
    got = part2.double().sum(0).cpu()
    scale_g = ref_dx.abs().sum(dim=(0, 2, 3)).max()
    assert ((got[0] - s_g).abs().max() / scale_g) < 2e-3 and ((got[1] - s_gx).abs().max() / (ref_dx.abs() * xs.abs()).sum(dim=(0, 2, 3)).max()) < 2e-3
    dw = ops.conv2d_wgrad_bn_bf16(gyd, yrawd, bcd, xd, find, (Co, k, k, Ci), s, p)
    check("bf16 fused wgrad", dw.cpu().permute(0, 3, 1, 2), w64.grad, 2e-3)


@pytest.mark.parametrize("geom,c64", [((4, 14, 14, 64, 64, 3, 1, 1), "0"), ((4, 14, 14, 64, 64, 3, 1, 1), "2"),
                                      ((4, 14, 14, 128, 64, 1, 1, 0), "0"), ((3, 15, 13, 64, 128, 3, 2, 1), "0"),
                                      ((6, 12, 12, 256, 64, 1, 1, 0), "0")])
def test_fused_bn_backward_large_mean_bf16(edrl, dev, geom, c64, switches):
    """bf16 counterpart of test_gpu_kernels.py::test_fused_bn_backward_large_mean (|mean| / sigma = 50 in the BatchNorm input): the
    epilogues of conv_bf16.hip (decision recomputed) and conv_c64_bf16.hip (sign bytes; c64 = "2" forces that kernel) emit
    (sum g, sum g*(x - mean)) from the UNROUNDED fp32 gradient; dgamma / dbeta vs fp64 autograd on the same bf16-rounded operands at
    a fixed 1e-4 (the gradient itself is an fp32 accumulation of bf16 products: 2e-5 class), d_raw (edrl_bn_draw_bf16, bf16
    storage of g and of the result) at two bf16 roundings."""
    from util import large_mean_case
    from edrl_amd import encoders as E
    ops, L = edrl.ops, edrl._lib
    P = L.ptr
    switches(EDRL_BF16_C64=c64)
    N, H, W, Ci, Co, k, s, p = geom
    x, w, dy, gamma, fc, (dg_ref, db_ref, dx_ref), pre, care = large_mean_case(*geom, seed=12, bf16=True)
    xdv, fcd, dyd = x.bfloat16().to(dev), fc.to(dev), dy.bfloat16().to(dev)
    wt = ops.permute_weight_bf16(w.to(dev))
    mask = None
    if c64 == "2":          # the weight-stationary kernel's epilogue takes the stored sign bytes
        keep = (pre > 0).permute(0, 2, 3, 1).reshape(N * H * W, Ci // 4, 4).to(torch.uint8)
        mask = (keep * torch.tensor([1, 2, 4, 8], dtype=torch.uint8)).sum(-1).to(torch.uint8).to(dev)
    gm, part, chunks = ops.conv2d_dgrad_bn_bf16(dyd, None, None, wt, (N, H, W, Ci), s, p, ep=(xdv, mask, fcd, True))
    bc, dgam, dbet = E._bcoef_from_partials(part, chunks, 2, N * H * W, gamma.to(dev), fcd)
    d_raw = torch.empty_like(gm)
    L.call("edrl_bn_draw_bf16", P(gm), P(xdv), P(bc), P(d_raw), N * H * W, Ci)
    torch.cuda.synchronize()
    check(f"bf16 large-mean dbeta {geom}", dbet.cpu(), db_ref, 1e-4)
    check(f"bf16 large-mean dgamma {geom}", dgam.cpu(), dg_ref, 1e-4)
    err = ((nchw(d_raw.double().cpu()) - dx_ref) * care).abs().max() / dx_ref.abs().max()
    print(f"[parity] bf16 large-mean d_raw {geom}: max-rel-err {err:.3e} (tol {2 ** -7:.1e})")
    assert err <= 2 ** -7


V3S_CASES = V3_CASES + [
    (3, 128, 14, 14, 128, 3, 1, 1),      # 128-channel layers: what the 256 x 256 core cannot take
    (2, 128, 13, 11, 384, 3, 2, 1),
    (5, 32, 9, 7, 128, 1, 1, 0),         # K = 32: one ring unit
    (2, 512, 28, 28, 128, 1, 1, 0),
    (1, 128, 5, 5, 128, 3, 1, 1),        # 25 rows: less than one tile, upper wave row entirely past the end
    (3, 128, 9, 9, 128, 1, 1, 0),        # 243 rows: the second tile's upper wave row is past the end (statistics re-base with nb = 0)
]


@pytest.mark.parametrize("case", V3S_CASES)
def test_conv_bf16_v3s_core_vs_fp64_and_128row_kernel(edrl, dev, case, switches):
    """The 128x128 small-tile LDS-DMA core (csrc/conv_bf16_v3s.hip: 4 waves, 64 KiB ring, two workgroups per CU), forced on
    (EDRL_BF16_V3S=2) wherever its geometry allows: forward with the BatchNorm chunk partials (one 128-row chunk per tile, the two
    wave rows re-based and combined through LDS), data gradient plain / accumulating / stride-2 parity classes -- against the
    fp64 convolution of the same bf16 operands at one bf16 ulp, the partials against fp64 sums over the output rows, and
    against the 128-row kernel (EDRL_BF16_V3S=0, EDRL_BF16_V3=0)."""
    ops = edrl.ops
    N, Ci, H, W, Co, k, s, p = case
    g = torch.Generator().manual_seed(35)
    x = torch.randn(N, H, W, Ci, generator=g).bfloat16()
    w = (torch.randn(Co, k, k, Ci, generator=g) * 0.1).bfloat16()
    xd = nchw(x.double()).requires_grad_(True)
    wd = w.double().permute(0, 3, 1, 2).contiguous().requires_grad_(True)
    y = F.conv2d(xd, wd, stride=s, padding=p)
    dy = torch.randn(y.shape, generator=g).bfloat16()
    y.backward(dy.double())
    outs = {}
    switches(EDRL_BF16_V3="0", EDRL_BF16_K64="0", EDRL_BF16_C64="0")
    for mode in ("0", "2"):
        switches(EDRL_BF16_V3S=mode)
        yh, part, chunks = ops.conv2d_fwd_bf16(x.to(dev), w.to(dev), s, p, stats=True)
        dxh = dxa = None
        if Ci % 128 == 0:
            wt = ops.permute_weight_bf16(w.float().to(dev))
            dyh = dy.permute(0, 2, 3, 1).contiguous().to(dev)
            dxh = ops.conv2d_dgrad_bf16(dyh, wt, (N, H, W, Ci), s, p)
            dxa = ops.conv2d_dgrad_bf16(dyh, wt, (N, H, W, Ci), s, p, out=dxh.clone(), accumulate=True)
        torch.cuda.synchronize()
        outs[mode] = (yh.float().cpu(), part.cpu(), dxh.float().cpu() if dxh is not None else None,
                      dxa.float().cpu() if dxa is not None else None)
    yv, pv, dv, dav = outs["2"]
    check(f"v3s conv_fwd{case}", nchw(yv), y, BF16_TOL)
    check(f"v3s vs 128-row conv_fwd{case}", yv, outs["0"][0], BF16_TOL)
    rows = y.detach().permute(0, 2, 3, 1).reshape(-1, Co)
    M = rows.shape[0]
    assert pv.shape[0] == (M + 127) // 128
    for c in range(pv.shape[0]):
        blk = rows[c * 128:min(M, (c + 1) * 128)]
        K = pv[c, 2].double()
        assert float((K - blk[0]).abs().max()) <= BF16_TOL * float(rows.abs().max()), "shift = the chunk's first row (fp32 accumulator)"
        d = blk - K
        sc = max(float(d.abs().sum(0).max()), 1e-6)
        assert float((pv[c, 0].double() - d.sum(0)).abs().max()) <= 1e-3 * sc, f"chunk {c} S1"
        sc2 = max(float((d * d).sum(0).max()), 1e-6)
        assert float((pv[c, 1].double() - (d * d).sum(0)).abs().max()) <= 1e-3 * sc2, f"chunk {c} S2"
    if dv is not None:
        check(f"v3s conv_dgrad{case}", nchw(dv), xd.grad, BF16_TOL)
        check(f"v3s conv_dgrad_accum{case}", nchw(dav), 2 * xd.grad, 2 * BF16_TOL)
        check(f"v3s vs 128-row conv_dgrad{case}", dv, outs["0"][2], BF16_TOL)


V3_EPI_CASES = [
    # N, Hi, Wi, Ci (dx channels, multiple of 256), Co, k, s, p, accumulate, sign bytes?
    (3, 14, 14, 256, 256, 3, 1, 1, False, True),
    (3, 14, 14, 256, 256, 3, 1, 1, False, False),      # decision recomputed from the raw tensor
    (2, 13, 11, 512, 256, 1, 1, 0, True, True),        # block-input gradient: accumulate, then mask
    (2, 13, 11, 256, 512, 3, 2, 1, False, True),       # stride-2 parity classes, ragged class sizes
    (1, 7, 7, 256, 256, 3, 1, 1, False, True),         # less than one 128-row chunk
    (5, 9, 7, 256, 1024, 1, 1, 0, True, False),
    (2, 12, 12, 512, 256, 1, 2, 0, True, True),        # 1x1 stride 2: three of the four classes have no tap (epilogue only)
    (20, 64, 64, 256, 256, 1, 1, 0, True, True),       # 320 tiles: several tiles per workgroup in the persistent form
    (18, 62, 62, 256, 256, 3, 1, 1, False, False),     # 271 tiles, ragged last tile, decision recomputed
]


@pytest.mark.parametrize("persist", ["2", "3", "0", "v3s"])      # persistent with LDS-staged rows / persistent register form / one tile per workgroup / the 128x128 small-tile core
@pytest.mark.parametrize("case", V3_EPI_CASES + [(3, 14, 14, 128, 128, 3, 1, 1, False, True), (2, 13, 11, 128, 512, 1, 1, 0, True, False),
                                                 (4, 28, 28, 384, 128, 3, 2, 1, False, True)])
def test_conv_dgrad_v3_epilogue_vs_fp64_and_128row_kernel(edrl, dev, case, persist, switches):
    """The BatchNorm-backward epilogue of the 256x256 LDS-DMA data-gradient core (conv_bf16_v3.hip EPI 1: accumulate, mask with
    the sign bytes / the recomputed ReLU decision of the BatchNorm below, (sum g, sum g*(x - mean)) per 128-row chunk), forced on
    far below its production sizes: the masked gradient against fp64 at one bf16 ulp (two with accumulate), every chunk's partial
    sums against fp64 sums over the STORED gradient (the kernel sums the values it stores before their final rounding: 2^-8 of
    sum |g|), and both against the 128-row kernel's epilogue (same chunk layout: one finalize serves either)."""
    ops = edrl.ops
    N, H, W, Ci, Co, k, s, p, accum, use_mask = case
    g = torch.Generator().manual_seed(31 + Ci + Co)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    w = (torch.randn(Co, k, k, Ci, generator=g) * 0.1).bfloat16()
    dy = torch.randn(N, Ho, Wo, Co, generator=g).bfloat16()
    xraw = torch.randn(N, H, W, Ci, generator=g).bfloat16()
    old = torch.randn(N, H, W, Ci, generator=g).bfloat16()
    fc = torch.empty(5, Ci)
    fc[0] = 0.3 * torch.randn(Ci, generator=g); fc[1].fill_(1.0); fc[2] = 0.5 + torch.rand(Ci, generator=g)
    fc[3] = 0.2 * torch.randn(Ci, generator=g); fc[4] = fc[3] - fc[0] * fc[2]
    pre = xraw.float() * fc[2] + fc[4]
    keep = pre > 0
    mask = None
    if use_mask:
        keep = torch.rand(N, H, W, Ci, generator=g) < 0.6
        mask = (keep.reshape(N * H * W, Ci // 4, 4).to(torch.uint8) * torch.tensor([1, 2, 4, 8], dtype=torch.uint8)).sum(-1).to(torch.uint8)
    xd = torch.zeros(N, Ci, H, W, dtype=torch.float64, requires_grad=True)
    F.conv2d(xd, w.double().permute(0, 3, 1, 2), stride=s, padding=p).backward(nchw(dy.double()))
    full = xd.grad.permute(0, 2, 3, 1) + (old.double() if accum else 0.0)
    ref = full * keep.double()
    care = torch.ones_like(keep) if use_mask else (pre.abs() > 1e-6 * pre.abs().max())
    wt = ops.permute_weight_bf16(w.float().to(dev))
    outs = {}
    switches(EDRL_BF16_V3_PERSIST=persist)      # "2" / "3": conv_bf16_v3p.hip (EPI 2 / EPI 1), "0": conv_bf16_v3.hip
    if persist == "v3s":                        # the small-tile core's epilogue (conv_bf16_v3s.hip EPI 1) against the 128-row kernel's
        switches(EDRL_BF16_V3_PERSIST="1", EDRL_BF16_V3="0")
    else:
        switches(EDRL_BF16_V3S="0")
    for mode in ("0", "2"):
        if persist == "v3s":
            switches(EDRL_BF16_V3S=mode)
        else:
            switches(EDRL_BF16_V3=mode)
        dst = old.clone().to(dev) if accum else None
        gm, part, chunks = ops.conv2d_dgrad_bn_bf16(dy.to(dev), None, None, wt, (N, H, W, Ci), s, p, out=dst, accumulate=accum,
                                                    ep=(xraw.to(dev), mask.to(dev) if use_mask else None, fc.to(dev), True))
        torch.cuda.synchronize()
        outs[mode] = (gm.float().cpu(), part.cpu(), chunks)
    gm, part, chunks = outs["2"]
    tol = BF16_TOL * (2 if accum else 1)
    err = ((gm.double() - ref) * care).abs().max() / ref.abs().max()
    print(f"[parity] v3 dgrad epilogue {case}: masked gradient max-rel-err {err:.3e} (tol {tol:.1e})")
    assert err <= tol
    e2 = ((gm - outs["0"][0]).double() * care).abs().max() / ref.abs().max()
    assert e2 <= tol, f"v3 vs 128-row kernel: {e2:.3e}"
    assert chunks == outs["0"][2] and part.shape == outs["0"][1].shape
    # chunk partial sums: rows of a parity class are numbered class by class, 128 per chunk
    rows = []
    for ph in range(s):
        for pw in range(s):
            h0, w0 = (ph - p) % s, (pw - p) % s
            sub = gm[:, h0::s, w0::s, :]
            subx = xraw[:, h0::s, w0::s, :].float()
            if sub.numel() == 0:
                continue
            rows.append((sub.reshape(-1, Ci).double(), subx.reshape(-1, Ci).double()))
    c = 0
    for gr, xr in rows:
        for r0 in range(0, gr.shape[0], 128):
            gb, xb = gr[r0:r0 + 128], xr[r0:r0 + 128] - fc[0].double()
            sc0 = max(float(gb.abs().sum(0).max()), 1e-6); sc1 = max(float((gb * xb).abs().sum(0).max()), 1e-6)
            assert float((part[c, 0].double() - gb.sum(0)).abs().max()) <= 2 ** -8 * sc0, f"chunk {c} sum g"
            assert float((part[c, 1].double() - (gb * xb).sum(0)).abs().max()) <= 2 ** -8 * sc1, f"chunk {c} sum g*(x-mean)"
            c += 1
    assert c == chunks
    tot = lambda t: t.double().sum(0)
    sc = ref.abs().sum(dim=(0, 1, 2)).max()
    assert float((tot(part)[0] - tot(outs["0"][1])[0]).abs().max() / sc) < 2e-3, "sum g: v3 vs 128-row epilogue"


@pytest.mark.parametrize("N,H,W", [(3, 32, 32), (2, 30, 34), (5, 8, 6), (1, 224, 224)])
def test_stem_bf16_mma_kernel_vs_fp64_and_fp32_mfma_stem(edrl, dev, N, H, W):
    """The 1-channel stem on the bf16 matrix pipe (edrl_stem_conv_s2d_bf16: 7x7/s2/p3 as a 4x4/p2 window over the space-to-depth
    image, image and weights rounded to bf16 in registers).  Against the fp64 convolution of the SAME bf16-rounded image and weights
    at one bf16 ulp of the output; BatchNorm chunk partials against sums over the kernel's own fp32 accumulators (via the stored
    output: 2^-8); and against the fp32-MFMA stem of round 3 (unrounded operands) within the operand rounding, 2^-6 of the max."""
    ops = edrl.ops
    g = torch.Generator().manual_seed(N * 7 + H)
    x = torch.rand(N, H, W, 1, generator=g)
    w = torch.randn(64, 7, 7, 1, generator=g) * 0.1
    xq, wq = x.bfloat16().double(), w.bfloat16().double()
    ref = F.conv2d(xq.permute(0, 3, 1, 2), wq.permute(0, 3, 1, 2), stride=2, padding=3)
    y, part, chunks, xs, folded = ops.stem_conv_fwd_bf16mma(x.to(dev), w.to(dev))
    torch.cuda.synchronize()
    assert folded and y.dtype == torch.bfloat16 and tuple(y.shape) == (N, H // 2, W // 2, 64)
    check(f"bf16-MMA stem {N}x{H}x{W}", nchw(y.float().cpu()), ref, BF16_TOL)
    y32, part32, chunks32, _, _ = ops.stem_conv_fwd_obf16(x.to(dev), w.to(dev))
    check(f"bf16-MMA stem vs fp32-MFMA stem {N}x{H}x{W}", y.float().cpu(), y32.float().cpu(), 2.0 ** -6)
    assert chunks == chunks32 == (N * (H // 2) * (W // 2) + 127) // 128
    rows = ref.permute(0, 2, 3, 1).reshape(-1, 64)
    M = rows.shape[0]
    pv = part.cpu()
    for c in range(0, chunks, max(1, chunks // 7)):
        blk = rows[c * 128:min(M, (c + 1) * 128)]
        K = pv[c, 2].double()
        assert float((K - blk[0]).abs().max()) <= BF16_TOL * float(rows.abs().max()), "shift = the chunk's first row"
        d = blk - K
        assert float((pv[c, 0].double() - d.sum(0)).abs().max()) <= 1e-3 * max(float(d.abs().sum(0).max()), 1e-6), f"chunk {c} S1"
        assert float((pv[c, 1].double() - (d * d).sum(0)).abs().max()) <= 1e-3 * max(float((d * d).sum(0).max()), 1e-6), f"chunk {c} S2"


@pytest.mark.parametrize("N,H,W", [(3, 14, 14), (2, 9, 7), (1, 8, 8), (5, 28, 28), (1, 3, 5), (12, 56, 56)])
def test_conv1x1_k64_bwd_one_pass_kernel_vs_fp64_and_two_kernel_path(edrl, dev, N, H, W, switches):
    """Backward of the expanding 1x1 layers of the first residual stage (64 -> 256) in ONE pass over (g, yraw)
    (csrc/conv1x1_bwd_bf16.hip): pixel counts off the 128-pixel tile, fewer tiles than workgroups, one tile only.  Against fp64 on
    the SAME storage-rounded operands (d_raw and the activation rounded to bf16 as the kernels hold them): weight gradient 2e-5 of
    its max (fp32 accumulation of bf16 products, ordered split reduction), masked data gradient one bf16 ulp, the summed partials
    (sum g2, sum g2*(x2 - mean)) 1e-3 of sum |.|; and against the two-kernel path of rounds 2-3 (fused weight gradient + fused data
    gradient with epilogue), which forms the same operands with the same arithmetic; run twice: bit-identical (deterministic)."""
    ops = edrl.ops
    Ci, Co = 64, 256
    g = torch.Generator().manual_seed(N * 131 + H)
    q = lambda t: t.bfloat16()
    gy = q(torch.randn(N, H, W, Co, generator=g))
    yraw = q(torch.randn(N, H, W, Co, generator=g) + 0.3)
    bc = torch.empty(4, Co); bc[0] = 0.5 + torch.rand(Co, generator=g); bc[1] = 0.05 * torch.randn(Co, generator=g)
    bc[2] = 0.05 * torch.randn(Co, generator=g); bc[3].zero_()
    x2 = q(torch.randn(N, H, W, Ci, generator=g) + 0.2)
    fin = torch.empty(5, Ci); fin[0] = 0.2 + 0.1 * torch.randn(Ci, generator=g); fin[1].fill_(1.0)
    fin[2] = 0.5 + torch.rand(Ci, generator=g); fin[3] = 0.2 * torch.randn(Ci, generator=g); fin[4] = fin[3] - fin[0] * fin[2]
    w = q(torch.randn(Co, 1, 1, Ci, generator=g) * 0.1)
    # fp64 reference on the storage-rounded operands
    # (the kernels' fused multiply-adds, one rounding each, emulated through fp64 so that no bf16 rounding boundary is crossed)
    inner = (bc[0].double() * gy.double() + bc[2].double()).float()
    d3 = q((bc[1].double() * yraw.double() + inner.double()).float()).double().reshape(-1, Co)
    pre = (x2.double() * fin[2].double() + fin[4].double()).float()
    a2 = q(torch.relu(pre)).double().reshape(-1, Ci)
    dw_ref = d3.t() @ a2
    keep = (pre > 0).reshape(-1, Ci)
    care = (pre.abs() > 1e-6 * pre.abs().max()).reshape(-1, Ci).double()
    g2_ref = (d3 @ w.double().reshape(Co, Ci)) * keep.double()
    dev_ = lambda t: t.to(dev)
    wt = ops.permute_weight_bf16(w.float().to(dev))
    assert ops.conv1x1_k64_bwd_ok_bf16(N, H, W, Ci, Co)
    dw, g2, part, chunks = ops.conv1x1_k64_bwd_bf16(dev_(gy), dev_(yraw), dev_(bc), dev_(x2), dev_(fin), wt)
    dw_b, g2_b, part_b, _ = ops.conv1x1_k64_bwd_bf16(dev_(gy), dev_(yraw), dev_(bc), dev_(x2), dev_(fin), wt)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw_b) and torch.equal(g2, g2_b) and torch.equal(part, part_b), "deterministic"
    check(f"one-pass 1x1 bwd dW {N}x{H}x{W}", dw.cpu().reshape(Co, Ci), dw_ref, 2e-5)
    err = ((g2.double().cpu().reshape(-1, Ci) - g2_ref) * care).abs().max() / g2_ref.abs().max()
    print(f"[parity] one-pass 1x1 bwd g2 {N}x{H}x{W}: max-rel-err {err:.3e} (tol {BF16_TOL:.1e})")
    assert err <= BF16_TOL
    xs = x2.double().reshape(-1, Ci) - fin[0].double()
    tot = part.double().sum(0).cpu()
    gm = g2_ref * care
    assert float((tot[0] - gm.sum(0)).abs().max()) <= 1e-3 * float(g2_ref.abs().sum(0).max()), "sum g2"
    assert float((tot[1] - (gm * xs).sum(0)).abs().max()) <= 1e-3 * float((g2_ref.abs() * xs.abs()).sum(0).max()), "sum g2*(x2-mean)"
    # the two-kernel path of rounds 2-3 on the same operands (where its fused fast paths exist: not on maps below ~4x4)
    if not ops.conv_fused_ok_bf16(N, H, W, Ci, Co, 1, 1, 0):
        return
    dw2 = ops.conv2d_wgrad_bn_bf16(dev_(gy), dev_(yraw), dev_(bc), dev_(x2), dev_(fin), (Co, 1, 1, Ci), 1, 0)
    g22, part2, _ = ops.conv2d_dgrad_bn_bf16(dev_(gy), dev_(yraw), dev_(bc), wt, (N, H, W, Ci), 1, 0, ep=(dev_(x2), None, dev_(fin), True))
    check("one-pass vs two-kernel dW", dw.cpu(), dw2.cpu(), 2e-5)
    e2 = ((g2.double() - g22.double()).cpu().reshape(-1, Ci) * care).abs().max() / g2_ref.abs().max()
    assert e2 <= BF16_TOL, f"one-pass vs two-kernel g2: {e2:.3e}"


@pytest.mark.parametrize("N,H,W", [(3, 32, 32), (2, 30, 34), (5, 8, 6), (2, 224, 224)])
def test_stem_wgrad_bf16_mma_kernel_vs_fp64_and_fp32_kernel(edrl, dev, N, H, W, switches):
    """Weight gradient of the 1-channel stem on the bf16 matrix pipe (edrl_stem_wgrad_s2d_bf16: d_raw bf16 x the space-to-depth fp32
    image rounded to bf16 in registers, 128-pixel tiles through swizzled LDS images, transposing fragment reads, ordered slab
    reduction).  Against the fp64 weight gradient of the 7x7/s2/p3 convolution on the SAME bf16-rounded image and gradient at 2e-5 of
    its max (fp32 accumulation of bf16 products); against the fp32 kernel it replaces (unrounded image) within the operand rounding,
    1e-2; bit-reproducible."""
    import os
    ops = edrl.ops
    g = torch.Generator().manual_seed(N * 11 + W)
    x = torch.rand(N, H, W, 1, generator=g)
    dy = torch.randn(N, H // 2, W // 2, 64, generator=g).bfloat16()
    wd = torch.zeros(64, 1, 7, 7, dtype=torch.float64, requires_grad=True)
    F.conv2d(x.bfloat16().double().permute(0, 3, 1, 2), wd, stride=2, padding=3).backward(nchw(dy.double()))
    ref = wd.grad.permute(0, 2, 3, 1)                                   # [64,7,7,1]
    xs = ops.space_to_depth2(x.to(dev))
    dw = ops.stem_conv_wgrad(dy.to(dev), xs, (64, 7, 7, 1), True)
    dw_b = ops.stem_conv_wgrad(dy.to(dev), xs, (64, 7, 7, 1), True)
    torch.cuda.synchronize()
    assert torch.equal(dw, dw_b), "deterministic"
    check(f"bf16-MMA stem wgrad {N}x{H}x{W}", dw.cpu(), ref, 2e-5)
    old = ops._STEM_WGRAD_MMA
    ops._STEM_WGRAD_MMA = False
    try:
        dw32 = ops.stem_conv_wgrad(dy.to(dev), xs, (64, 7, 7, 1), True)
    finally:
        ops._STEM_WGRAD_MMA = old
    check(f"bf16-MMA stem wgrad vs fp32 kernel {N}x{H}x{W}", dw.cpu(), dw32.cpu(), 1e-2)
