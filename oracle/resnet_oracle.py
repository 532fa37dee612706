"""CPU oracle for the build-owned encoders — TEST INFRASTRUCTURE ONLY.

The reference's encoder sources are absent (`Models/` is not in the reference repository; call
sites fusion_net.py:796,799,884-885), so parity for E1-E3 is *unpinned by the reference*: the oracle
is the standard composition of torch CPU ops (F.conv2d, F.batch_norm(training=True), F.relu,
F.max_pool2d, adaptive average pooling, F.linear) in the He et al. v1.5 ResNet topology, as
SURVEY.md §8(c) prescribes.  It consumes the product's parameters by name (conv weights are stored
[Co,KH,KW,Ci] by the product and permuted to torch's [Co,Ci,KH,KW] here).
"""
import torch
import torch.nn.functional as F


def _bn(x, sd, name, train=True):
    rm = sd[name + ".running_mean"]
    rv = sd[name + ".running_var"]
    return F.batch_norm(x, rm, rv, sd[name + ".weight"], sd[name + ".bias"], train, 0.1, 1e-5)


def _conv(x, sd, name, stride, pad):
    w = sd[name + ".weight"].permute(0, 3, 1, 2)
    return F.conv2d(x, w[:, :x.shape[1]], stride=stride, padding=pad)


def _relu(x, pins, key):
    """ReLU, or — when `pins` holds a boolean mask for `key` — multiplication by that mask: the discrete decision is
    taken from the product instead of from this precision's own sign test (an fp32 and an fp64 pipeline disagree on
    the sign of a pre-activation within round-off of zero; pinning removes that one ill-conditioned bit per element
    from a gradient comparison).  Disagreements between the pinned mask and the oracle's own sign are counted."""
    if pins is None or key not in pins:
        return F.relu(x)
    m = pins[key]
    pins.setdefault("_flips", {})[key] = int(((x.detach() > 0) != m).sum())
    return x * m.to(x.dtype)


def _maxpool(x, pins):
    if pins is None or "maxpool" not in pins:
        return F.max_pool2d(x, 3, 2, 1)
    idx = pins["maxpool"].long()                       # [N,C,Ho,Wo], kh*3+kw of the product's arg-max
    N, C, H, W = x.shape
    Ho, Wo = idx.shape[2], idx.shape[3]
    xp = F.pad(x, (1, 1, 1, 1))
    rows = 2 * torch.arange(Ho).view(1, 1, Ho, 1) + idx // 3
    cols = 2 * torch.arange(Wo).view(1, 1, 1, Wo) + idx % 3
    out = xp.flatten(2).gather(2, (rows * (W + 2) + cols).flatten(2)).view(N, C, Ho, Wo)
    own = F.max_pool2d(x.detach(), 3, 2, 1)
    pins.setdefault("_flips", {})["maxpool"] = int((own != out.detach()).sum())
    return out


def trunk_forward(x, sd, kind, blocks, train=True, pins=None):
    """x NCHW -> feature map NCHW. `sd`: dict name -> tensor (parameters may require grad; running
    stats are updated in place like nn.BatchNorm2d).  `pins` (optional): {conv name: bool ReLU mask NCHW, "maxpool":
    arg-max tap index NCHW} taken from the product (see _relu)."""
    x = _relu(_bn(_conv(x, sd, "conv1", 2, 3), sd, "bn1", train), pins, "conv1")
    x = _maxpool(x, pins)
    for blk in blocks:
        pre, s = blk["name"], blk["stride"]
        idn = x
        if kind == "bottleneck":
            o = _relu(_bn(_conv(x, sd, pre + ".conv1", 1, 0), sd, pre + ".bn1", train), pins, pre + ".conv1")
            o = _relu(_bn(_conv(o, sd, pre + ".conv2", s, 1), sd, pre + ".bn2", train), pins, pre + ".conv2")
            o = _bn(_conv(o, sd, pre + ".conv3", 1, 0), sd, pre + ".bn3", train)
            last = pre + ".conv3"
        else:
            o = _relu(_bn(_conv(x, sd, pre + ".conv1", s, 1), sd, pre + ".bn1", train), pins, pre + ".conv1")
            o = _bn(_conv(o, sd, pre + ".conv2", 1, 1), sd, pre + ".bn2", train)
            last = pre + ".conv2"
        if blk["downsample"]:
            idn = _bn(_conv(x, sd, pre + ".downsample.0", s, 0), sd, pre + ".downsample.1", train)
        x = _relu(o + idn, pins, last)
    return x


def _q16(t):
    """Round to bf16 storage (straight-through for autograd): the value the product keeps in HBM."""
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def conv_bn_bf16_op(x, w, gamma, beta, stride, pad, relu, residual=None):
    """One conv -> BatchNorm(train) -> (+residual) -> (ReLU) building block of the bf16 trunk, NCHW, working precision
    of the inputs (fp64 in the tests), with the product's storage roundings (encoders._conv_bn_fwd_bf16): bf16 weights,
    batch statistics from the UNROUNDED accumulators, conv output rounded to bf16 before normalisation, bf16 result.
    w is [Co,KH,KW,Ci].  -> (raw_q, out_q, mean, var)."""
    acc = F.conv2d(x, _q16(w).permute(0, 3, 1, 2)[:, :x.shape[1]], stride=stride, padding=pad)
    mean = acc.mean(dim=(0, 2, 3), keepdim=True)
    var = acc.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    raw = _q16(acc)
    y = (raw - mean) * torch.rsqrt(var + 1e-5) * gamma.view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return raw, _q16(y), mean.flatten(), var.flatten()


def trunk_forward_bf16(x, sd, kind, blocks, stem_q16=False):
    """Storage-aware restatement of the product's bf16 trunk (encoders._TrunkBf16Fn, SURVEY.md §8a rows E1/E2 at
    C2/C4): same topology as trunk_forward, with a bf16 rounding at exactly the tensors the product stores in bf16
    (conv_bn_bf16_op).  The stem conv (fp32 image and weights in the product) and all BatchNorm arithmetic stay in the working
    precision; the stem's raw output is stored as bf16 like every other layer's (statistics from the unrounded values, as the
    product takes them from the fp32 accumulators).  Training-mode batch statistics only (running statistics are not touched).
    Autograd sees straight-through roundings.  stem_q16: the 1-channel (OCT) stem of the product runs on the bf16 matrix pipe
    (encoders._KBF16.stem_fwd, EDRL_BF16_STEM_MMA): image and stem weights are rounded to bf16 on the way into the MFMA."""
    def cb(x, cname, bname, stride, pad, relu, residual=None):
        return conv_bn_bf16_op(x, sd[cname + ".weight"], sd[bname + ".weight"], sd[bname + ".bias"], stride, pad, relu,
                               residual)[1]

    if stem_q16:
        a = F.conv2d(_q16(x), _q16(sd["conv1.weight"]).permute(0, 3, 1, 2)[:, :x.shape[1]], stride=2, padding=3)
    else:
        a = _conv(x, sd, "conv1", 2, 3)
    mean = a.mean(dim=(0, 2, 3), keepdim=True)
    var = a.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
    a = (_q16(a) - mean) * torch.rsqrt(var + 1e-5) * sd["bn1.weight"].view(1, -1, 1, 1) + sd["bn1.bias"].view(1, -1, 1, 1)
    x = F.max_pool2d(_q16(F.relu(a)), 3, 2, 1)
    for blk in blocks:
        pre, s = blk["name"], blk["stride"]
        idn = cb(x, pre + ".downsample.0", pre + ".downsample.1", s, 0, False) if blk["downsample"] else x
        if kind == "bottleneck":
            o = cb(x, pre + ".conv1", pre + ".bn1", 1, 0, True)
            o = cb(o, pre + ".conv2", pre + ".bn2", s, 1, True)
            x = cb(o, pre + ".conv3", pre + ".bn3", 1, 0, True, idn)
        else:
            o = cb(x, pre + ".conv1", pre + ".bn1", s, 1, True)
            x = cb(o, pre + ".conv2", pre + ".bn2", 1, 1, True, idn)
    return x


def fundus_encoder_forward(x, sd, kind, blocks, proj_w, proj_b, train=True, pins=None):
    """[B,3,H,W] -> (tokens [B,N2,D], pooled).  pins: see trunk_forward."""
    f = trunk_forward(x, sd, kind, blocks, train, pins=pins)
    B, C, h, w = f.shape
    tok = f.permute(0, 2, 3, 1).reshape(B, h * w, C)
    tokens = F.linear(tok, proj_w, proj_b)
    return tokens, tokens.mean(1)


def oct_encoder_forward(x, sd, kind, blocks, proj_w, proj_b, train=True, pins=None):
    """[B,1,S,H,W] -> (tokens [B,S,D], pooled).  pins: see trunk_forward."""
    B, C, S, H, W = x.shape
    f = trunk_forward(x.reshape(B * S, 1, H, W), sd, kind, blocks, train, pins=pins)
    pooled = f.mean(dim=(2, 3)).reshape(B, S, -1)
    tokens = F.linear(pooled, proj_w, proj_b)
    return tokens, tokens.mean(1)


def trunk_state(trunk, dtype=torch.float64, requires_grad=True):
    """Copy a product ResNetTrunk's parameters/buffers to CPU tensors keyed by dotted names."""
    sd = {}
    for n, p in trunk.named_parameters():
        sd[n] = p.detach().cpu().to(dtype).requires_grad_(requires_grad)
    for n, b in trunk.named_buffers():
        n = n.replace("__", ".")
        sd[n] = b.detach().cpu().to(dtype) if b.dtype.is_floating_point else b.detach().cpu().clone()
    return sd


# ---------------------------------------------------------------------------------------------- 3-D-conv OCT trunk
def _w3d(w, k, ci):
    """product layout [Co,KH,KW,CK] (CK = KD*Ci padded) -> torch [Co,Ci,KD,KH,KW]"""
    co = w.shape[0]
    return w[..., : k * ci].reshape(co, k, k, k, ci).permute(0, 4, 3, 1, 2)


def trunk3d_forward(x, sd, n_blocks_per_layer, train=True):
    """Oracle of encoders3d.ResNet3DTrunk (parity unpinned by the reference: no source): torch-CPU
    F.conv3d / F.batch_norm(training=True) / F.relu / F.max_pool3d in the MedicalNet-style basic-block topology.
    x [N,1,D,H,W]; sd: name -> tensor from trunk3d_state(); -> [N,512,d,h,w]."""
    def conv(x, name, k, stride, pad):
        return F.conv3d(x, _w3d(sd[name + ".weight"], k, x.shape[1]), stride=stride, padding=pad)

    def bn(x, name):
        return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"],
                            sd[name + ".bias"], train, 0.1, 1e-5)

    x = F.max_pool3d(F.relu(bn(conv(x, "conv1", 7, 2, 3), "bn1")), 3, 2, 1)
    bi, ci = 0, 64
    for li, (co, n) in enumerate(zip([64, 128, 256, 512], n_blocks_per_layer)):
        for b in range(n):
            stride = 2 if (b == 0 and li > 0) else 1
            pre = f"blocks.{bi}"
            idn = x
            if stride != 1 or ci != co:
                idn = bn(conv(x, pre + ".down.0", 1, stride, 0), pre + ".down.1")
            o = F.relu(bn(conv(x, pre + ".conv1", 3, stride, 1), pre + ".bn1"))
            x = F.relu(bn(conv(o, pre + ".conv2", 3, 1, 1), pre + ".bn2") + idn)
            bi += 1
            ci = co
    return x


def trunk3d_state(trunk, dtype=torch.float64, requires_grad=True):
    sd = {}
    for n, p in trunk.named_parameters():
        sd[n] = p.detach().cpu().to(dtype).requires_grad_(requires_grad)
    for n, b in trunk.named_buffers():
        sd[n] = b.detach().cpu().to(dtype) if b.dtype.is_floating_point else b.detach().cpu().clone()
    return sd


def pins_from_capture(cap):
    """Discrete decisions of one product trunk pass (ResNetTrunk._capture record) in the form trunk_forward(pins=...) takes:
    ReLU sign bits per conv unit (NCHW bool) and the max-pool arg-max taps."""
    pins = {}
    for name, rec in cap.items():
        if name.startswith("bwd:") or name == "maxpool" or not rec["relu"]:
            continue
        pins[name] = (rec["out"].detach().permute(0, 3, 1, 2).cpu() > 0)
    pins["maxpool"] = cap["maxpool"]["idx"].permute(0, 3, 1, 2).cpu()
    return pins
