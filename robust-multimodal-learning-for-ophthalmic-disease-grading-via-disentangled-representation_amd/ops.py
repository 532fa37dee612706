"""Operator layer: torch.autograd.Function wrappers over the C-ABI HIP launchers.

Every forward and backward below runs hand-written gfx950 kernels from libedrl_hip.so on torch's
current stream.  torch is used only for device memory (torch.empty from the caching allocator),
views/concatenation and autograd bookkeeping.  There is no eager/CPU fallback: tensors that are
not fp32 CUDA tensors are rejected.
"""
import os

import torch

from . import _lib as L

P = L.ptr

EW_RELU, EW_RELU_BWD, EW_AXPBY, EW_MUL, EW_SCALE, EW_SOFTPLUS, EW_SOFTPLUS_BWD, EW_MASKED_BWD, \
    EW_ADD_RELU, EW_SCALE_BY_PTR, EW_FILL, EW_LERP_BY_PTR = range(12)
FLAG_RELU, FLAG_ACCUM = 1, 2


def _chk(t, name="tensor", contiguous=True):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError(f"{name}: expected a float32 CUDA tensor (the EDRL hot path is HIP-only), "
                           f"got {type(t).__name__} {getattr(t, 'dtype', None)} {getattr(t, 'device', None)}")
    if contiguous and not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor, got strides {t.stride()}")
    return t


def _rows2d(x):
    """View x [..., D] as [rows, D] with unit inner stride (row stride may exceed D)."""
    if x.stride(-1) != 1:
        x = x.contiguous()
    x2 = x.reshape(-1, x.shape[-1])
    if x2.stride(-1) != 1 or (x2.shape[0] > 1 and x2.stride(0) < x2.shape[1]):
        x2 = x2.contiguous()
    return x2


def _ld(x2):
    return x2.stride(0) if x2.shape[0] > 1 else max(x2.stride(0), x2.shape[1])


# ------------------------------------------------------------------ live kernel timing (bench.py roofline leg)
class KernelTimer:
    """HIP-event bracketing of launches on the stream they run on.
    kind -> [algorithmic flops, [(start_event, end_event), ...], kernel launches, algorithmic bytes]."""

    def __init__(self):
        self.records = {}

    PEAK_BW = 8.0e12                                    # HBM3E, MI355X_MICROARCH.md
    PEAK_F32, PEAK_BF16 = 157.3e12, 2.5e15              # dense MFMA peaks
    _peak_f32 = None

    def peak_f32(self):
        """Matrix-pipe ceiling for algorithmic fp32 FLOPs of the loaded library: bf16 dense peak / 6 when it forms fp32 products as
        bf16x3 splits (edrl_f32_contraction_split), the fp32 MFMA's peak otherwise."""
        if KernelTimer._peak_f32 is None:
            KernelTimer._peak_f32 = 2516.6e12 / 6.0 if L.lib().fn["edrl_f32_contraction_split"]() else self.PEAK_F32
        return KernelTimer._peak_f32

    def add(self, kind, flops, e0, e1, kernels=1, nbytes=0.0):
        r = self.records.setdefault(kind, [0.0, [], 0, 0.0, 0.0])
        r[0] += flops
        # speed-of-light time of THIS call: whichever of its algorithmic flops / bytes binds
        t_f = flops / (self.PEAK_BF16 if kind.endswith("bf16") else self.peak_f32())
        t_b = nbytes / self.PEAK_BW
        r[1].append((e0, e1, t_b > t_f, flops, nbytes))
        r[2] += kernels
        r[3] += nbytes
        r[4] += max(t_f, t_b)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for kind, (flops, evs, kernels, nbytes, bound_s) in self.records.items():
            ms = 0.0
            split = {False: [0, 0.0, 0.0, 0.0], True: [0, 0.0, 0.0, 0.0]}        # hbm-bound? -> calls, ms, flops, bytes
            for a, b, hbm, fl, by in evs:
                t = a.elapsed_time(b)
                ms += t
                c = split[hbm]
                c[0] += 1; c[1] += t; c[2] += fl; c[3] += by
            out[kind] = {"launches": kernels, "calls": len(evs), "flops": flops, "ms": ms,
                         "tflops": flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, "bound_ms": bound_s * 1e3}
            if flops > 0 and nbytes > 0:     # the family split by which resource binds each call (ridge = peak flops / 8 TB/s)
                m, h = split[False], split[True]
                out[kind]["by_bound"] = {
                    "mfma": {"calls": m[0], "ms": m[1], "tflops": m[2] / (m[1] * 1e-3) / 1e12 if m[1] > 0 else 0.0},
                    "hbm": {"calls": h[0], "ms": h[1], "GBps": h[3] / (h[1] * 1e-3) / 1e9 if h[1] > 0 else 0.0}}
            if nbytes:
                out[kind]["bytes"] = nbytes
                out[kind]["GBps"] = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return out


_timer = None


def set_timer(t):
    global _timer
    _timer = t


def _launch_timed(kind, flops, name, *args, kernels=1, nbytes=0.0):
    """`kernels`: MFMA kernels the C-ABI call issues (a strided data gradient issues one per non-empty parity class),
    so that the timer's launch count is the one rocprofv3 sees.  `nbytes`: algorithmic HBM bytes of the call (every operand
    tensor read once, every result written once; weights included)."""
    if _timer is None:
        L.trace_note(kind=kind, flops=flops, nbytes=nbytes)      # (no-op unless the call tracer is on: _lib.trace_begin)
        L.call(name, *args)
        return
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    k0 = L.query("edrl_gather_launch_count")
    e0.record()
    L.call(name, *args)
    e1.record()
    k1 = L.query("edrl_gather_launch_count")
    _timer.add(kind, flops, e0, e1, (k1 - k0) if k1 > k0 else kernels, nbytes)     # fp32 gather family: the launches actually issued


def call_timed_bytes(kind, nbytes, name, *args, kernels=1):
    """An HBM-bound launcher bracketed like _launch_timed; `nbytes` = its algorithmic bytes (tensor reads + writes)."""
    if _timer is None:
        L.trace_note(kind=kind, flops=0.0, nbytes=float(nbytes))
        L.call(name, *args)
        return
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    L.call(name, *args)
    e1.record()
    _timer.add(kind, 0.0, e0, e1, kernels, float(nbytes))


def _dgrad_kernels(Hi, Wi, KH, KW, stride, pad, accumulate):
    """Number of gather kernels edrl_conv2d_nhwc_dgrad_* launches (mirrors its parity-class loop)."""
    n = 0
    for ph in range(stride):
        for pw in range(stride):
            h0, w0 = (ph - pad) % stride, (pw - pad) % stride
            if h0 >= Hi or w0 >= Wi:
                continue
            khs = (KH - ph + stride - 1) // stride if ph < KH else 0
            kws = (KW - pw + stride - 1) // stride if pw < KW else 0
            if khs * kws == 0 and accumulate:
                continue
            n += 1
    return n


# ------------------------------------------------------------------ raw (non-autograd) launch helpers
def ew(op, a, b=None, c=None, out=None, alpha=1.0, beta=1.0):
    a = a.contiguous()
    if out is None:
        out = torch.empty_like(a)
    L.call("edrl_ew_f32", op, a.numel(), P(a), P(b), P(c), P(out), float(alpha), float(beta))
    return out


def conv2d_fwd(x, w, bias=None, mul=None, stride=1, pad=0, relu=False, out=None, accumulate=False, out_hw=None,
               alg_flops=None):
    """x [N,Hi,Wi,Ci] NHWC, w [Co,KH,KW,Ci] -> y [N,Ho,Wo,Co].  out_hw overrides (Ho, Wo) (asymmetric padding: taps
    beyond the bottom/right edge read zeros); alg_flops overrides the algorithmic FLOP count given to the kernel timer."""
    N, Hi, Wi, Ci = x.shape
    Co, KH, KW, _ = w.shape
    Ho = (Hi + 2 * pad - KH) // stride + 1
    Wo = (Wi + 2 * pad - KW) // stride + 1
    if out_hw is not None:
        Ho, Wo = out_hw
    if out is None:
        out = torch.empty((N, Ho, Wo, Co), device=x.device, dtype=torch.float32)
    flags = (FLAG_RELU if relu else 0) | (FLAG_ACCUM if accumulate else 0)
    _launch_timed("conv_gather", alg_flops or 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_fwd_f32", P(x), P(w), P(bias),
                  P(mul), P(out), N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, Ci, Co, Co, flags,
                  nbytes=4.0 * (x.numel() + w.numel() + out.numel() * (2 if accumulate else 1)))
    return out


def conv2d_fwd_stats(x, w, stat_shift, stride=1, pad=0):
    """Conv forward with the BatchNorm chunk partials of its output produced by the epilogue.
    -> (y [N,Ho,Wo,Co], part [chunks][3][Co], chunks).  Needs Ci % 16 == 0."""
    N, Hi, Wi, Ci = x.shape
    Co, KH, KW, _ = w.shape
    Ho = (Hi + 2 * pad - KH) // stride + 1
    Wo = (Wi + 2 * pad - KW) // stride + 1
    out = torch.empty((N, Ho, Wo, Co), device=x.device, dtype=torch.float32)
    chunks = L.query("edrl_conv_stats_chunks", N, Ho, Wo)
    part = torch.empty((chunks, 3, Co), device=x.device, dtype=torch.float32)
    _launch_timed("conv_gather", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_fwd_stats_f32", P(x), P(w), P(out),
                  P(stat_shift), P(part), part.numel() * 4, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad,
                  nbytes=4.0 * (x.numel() + w.numel() + out.numel()))
    return out, part, chunks


def space_to_depth2(x):
    """[N,H,W,C] -> [N,H/2,W/2,4C], channel order (ph, pw, c)."""
    N, H, W, C = x.shape
    y = torch.empty((N, H // 2, W // 2, 4 * C), device=x.device, dtype=torch.float32)
    L.call("edrl_space_to_depth2_f32", P(x), P(y), N, H, W, C)
    return y


def stem_weight_fold(w, inverse=False):
    """7x7 stem weight [Co,7,7,C] <-> its 4x4 space-to-depth form [Co,4,4,4C] (inverse: gather back, for gradients)."""
    Co = w.shape[0]
    if inverse:
        C = w.shape[3] // 4
        out = torch.empty((Co, 7, 7, C), device=w.device, dtype=torch.float32)
    else:
        C = w.shape[3]
        out = torch.empty((Co, 4, 4, 4 * C), device=w.device, dtype=torch.float32)
    L.call("edrl_stem_weight_fold_f32", P(w.contiguous()), P(out), Co, C, 1 if inverse else 0)
    return out


def stem_conv_fwd(x, w):
    """The 7x7/s2/p3 stem conv.  Even H, W: 4x4/s1 conv on the space-to-depth image (vector MFMA path, K = 64*C/…);
    -> (y [N,H/2,W/2,Co], the tensor to keep for the weight gradient, folded flag)."""
    N, H, W, C = x.shape
    Co = w.shape[0]
    if (H % 2 == 0) and (W % 2 == 0) and tuple(w.shape[1:3]) == (7, 7):
        xs = space_to_depth2(x)
        y = conv2d_fwd(xs, stem_weight_fold(w), stride=1, pad=2, out_hw=(H // 2, W // 2),
                       alg_flops=2.0 * N * (H // 2) * (W // 2) * Co * 49 * C)
        return y, xs, True
    return conv2d_fwd(x, w, stride=2, pad=3), x, False


def stem_conv_fwd_obf16(x, w):
    """The stem conv of the bf16 trunk: fp32 image and weights, fp32 MFMA, result stored as bf16, BatchNorm chunk partials
    from the fp32 accumulators.  -> (y bf16 [N,H/2,W/2,Co], part [chunks][3][Co], chunks, tensor kept for the weight gradient,
    folded flag); same two geometries as stem_conv_fwd."""
    N, H, W, C = x.shape
    Co = w.shape[0]
    if (H % 2 == 0) and (W % 2 == 0) and tuple(w.shape[1:3]) == (7, 7):
        xin, wk, stride, pad, Ho, Wo, folded = space_to_depth2(x), stem_weight_fold(w), 1, 2, H // 2, W // 2, True
    else:
        KH, KW = w.shape[1], w.shape[2]
        xin, wk, stride, pad, folded = x, w, 2, 3, False
        Ho, Wo = (H + 6 - KH) // 2 + 1, (W + 6 - KW) // 2 + 1
    _, Hi, Wi, Ci = xin.shape
    KH, KW = wk.shape[1], wk.shape[2]
    out = torch.empty((N, Ho, Wo, Co), device=x.device, dtype=torch.bfloat16)
    chunks = L.query("edrl_conv_stats_chunks", N, Ho, Wo)
    part = torch.empty((chunks, 3, Co), device=x.device, dtype=torch.float32)
    _launch_timed("conv_gather", 2.0 * N * Ho * Wo * Co * 49 * C if folded else 2.0 * N * Ho * Wo * Co * KH * KW * Ci,
                  "edrl_conv2d_nhwc_fwd_stats_f32_obf16", P(xin), P(wk), P(out), P(part), part.numel() * 4, N, Hi, Wi, Ci, Ho, Wo,
                  Co, KH, KW, stride, pad, nbytes=4.0 * (xin.numel() + wk.numel()) + 2.0 * out.numel())
    return out, part, chunks, xin, folded


_STEM_WGRAD_MMA = os.environ.get("EDRL_BF16_STEM_WGRAD_MMA", "1") != "0"     # 0: the fp32 weight-gradient kernel with bf16 dy widened on load


def stem_conv_fwd_bf16mma(x, w):
    """The stem of the 1-channel bf16 trunk on the bf16 matrix pipe (edrl_stem_conv_s2d_bf16): x fp32 [N,H,W,1] with even H, W,
    w fp32 [64,7,7,1].  -> (y bf16 [N,H/2,W/2,64], part, chunks, space-to-depth image kept for the weight gradient, True)."""
    N, H, W, C = x.shape
    assert C == 1 and w.shape[0] == 64 and tuple(w.shape[1:3]) == (7, 7) and H % 2 == 0 and W % 2 == 0
    xs = space_to_depth2(x)
    wk = to_bf16(stem_weight_fold(w))                      # [64,4,4,4] -> bf16 [64][64]
    Hs, Ws = H // 2, W // 2
    out = torch.empty((N, Hs, Ws, 64), device=x.device, dtype=torch.bfloat16)
    chunks = L.query("edrl_conv_stats_chunks", N, Hs, Ws)
    part = torch.empty((chunks, 3, 64), device=x.device, dtype=torch.float32)
    _launch_timed("conv_gather_bf16", 2.0 * N * Hs * Ws * 64 * 49, "edrl_stem_conv_s2d_bf16", P(xs), P(wk), P(out), P(part),
                  part.numel() * 4, N, Hs, Ws, nbytes=4.0 * xs.numel() + 2.0 * (wk.numel() + out.numel()))
    return out, part, chunks, xs, True


def stem_conv_wgrad(dy, x_saved, w_shape, folded):
    """dy fp32, or bf16 (the bf16 trunk's stem: fp32 x, dy widened on load)."""
    Co, _, _, C = w_shape
    if (folded and _STEM_WGRAD_MMA and dy.dtype == torch.bfloat16 and C == 1 and Co == 64 and x_saved.shape[-1] == 4
            and tuple(w_shape[1:3]) == (7, 7)):
        # 1-channel stem of the bf16 trunk: bf16 MFMA streaming kernel (image rounded to bf16 in registers)
        N, Hs, Ws, _ = dy.shape
        nbytes = L.query("edrl_stem_wgrad_s2d_bf16_workspace_bytes", N, Hs, Ws)
        ws = torch.empty(nbytes // 4, device=dy.device, dtype=torch.float32)
        dw8 = torch.empty((64, 4, 4, 4), device=dy.device, dtype=torch.float32)
        _launch_timed("conv_wgrad_bf16", 2.0 * N * Hs * Ws * 64 * 49, "edrl_stem_wgrad_s2d_bf16", P(dy), P(x_saved), P(dw8), P(ws), nbytes,
                      N, Hs, Ws, kernels=2, nbytes=2.0 * dy.numel() + 4.0 * (x_saved.numel() + dw8.numel()))
        return stem_weight_fold(dw8, inverse=True)
    if not folded:
        return conv2d_wgrad(dy, x_saved, tuple(w_shape), 2, 3)
    Co, _, _, C = w_shape
    N, Ho, Wo, _ = dy.shape
    dw8 = conv2d_wgrad(dy, x_saved, (Co, 4, 4, 4 * C), 1, 2, alg_flops=2.0 * N * Ho * Wo * Co * 49 * C)
    return stem_weight_fold(dw8, inverse=True)


def permute_weight(w):
    """[Co,KH,KW,Ci] -> [Ci,KH,KW,Co] (or [out,in] -> [in,out])."""
    if w.dim() == 2:
        A, B, C = w.shape[0], 1, w.shape[1]
        out = torch.empty((C, A), device=w.device, dtype=torch.float32)
    else:
        A, B, C = w.shape[0], w.shape[1] * w.shape[2], w.shape[3]
        out = torch.empty((C, w.shape[1], w.shape[2], A), device=w.device, dtype=torch.float32)
    L.call("edrl_permute_weight_f32", P(w), P(out), A, B, C)
    return out


def conv2d_dgrad(dy, wt, x_shape, stride=1, pad=0, out=None, accumulate=False):
    """dy [N,Ho,Wo,Co], wt [Ci,KH,KW,Co] -> dx [N,Hi,Wi,Ci]."""
    N, Hi, Wi, Ci = x_shape
    _, Ho, Wo, Co = dy.shape
    KH, KW = wt.shape[1], wt.shape[2]
    if out is None:
        out = torch.empty((N, Hi, Wi, Ci), device=dy.device, dtype=torch.float32)
    _launch_timed("conv_gather", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_dgrad_f32", P(dy), P(wt),
                  P(out), N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, Co, Ci, FLAG_ACCUM if accumulate else 0,
                  kernels=_dgrad_kernels(Hi, Wi, KH, KW, stride, pad, accumulate),
                  nbytes=4.0 * (dy.numel() + wt.numel() + out.numel() * (2 if accumulate else 1)))
    return out


def conv2d_wgrad(dy, x, w_shape, stride=1, pad=0, out=None, accumulate=False, alg_flops=None):
    N, Hi, Wi, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    KH, KW = w_shape[1], w_shape[2]
    if out is None:
        out = torch.empty(w_shape, device=dy.device, dtype=torch.float32)
        accumulate = False
    nbytes = L.query("edrl_conv2d_nhwc_wgrad_workspace_bytes", N, Ho, Wo, Co, Ci, KH, KW)
    ws = torch.empty(nbytes // 4, device=dy.device, dtype=torch.float32)
    if dy.dtype == torch.bfloat16:      # fp32 x, bf16 dy (widened exactly on load): the bf16 trunk's stem
        if L.query("edrl_conv2d_nhwc_wgrad_f32_dybf16_ok", N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW):
            _launch_timed("conv_wgrad", alg_flops or 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_wgrad_f32_dybf16", P(dy),
                          P(x), P(out), P(ws), nbytes, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, 1 if accumulate else 0,
                          nbytes=2.0 * dy.numel() + 4.0 * (x.numel() + out.numel()))
            return out
        dy = to_f32(dy)                 # geometry outside the buffer-load path: the same values through the fp32 kernel
    _launch_timed("conv_wgrad", alg_flops or 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_wgrad_f32", P(dy), P(x),
                  P(out), P(ws), nbytes, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, Co, Ci, 1 if accumulate else 0,
                  nbytes=4.0 * (dy.numel() + x.numel() + out.numel()))
    return out


# ---- convolutions with the neighbouring BatchNorm passes folded into their operand loads / epilogues (include/edrl_hip.h:
# fcoef [5][C] = {mean, rstd, scale, shift, shift2 = shift - mean*scale}, bcoef [4][C] = {A, nK2, C2, mean}; include/edrl_hip.h)
def conv_fused_ok(N, Hi, Wi, Ci, Co, KH, stride, pad):
    Ho = (Hi + 2 * pad - KH) // stride + 1
    Wo = (Wi + 2 * pad - KH) // stride + 1
    return bool(L.query("edrl_conv2d_fused_ok_f32", N, Hi, Wi, Ci, Ho, Wo, Co, KH, KH, stride, pad))


def conv2d_fwd_bnin_stats(x_raw, in_fcoef, w, stride=1, pad=0):
    """y = conv(relu(bn(x_raw; in_fcoef)), w) + the BatchNorm chunk partials of y.  -> (y, part, chunks)."""
    N, Hi, Wi, Ci = x_raw.shape
    Co, KH, KW, _ = w.shape
    Ho = (Hi + 2 * pad - KH) // stride + 1
    Wo = (Wi + 2 * pad - KW) // stride + 1
    out = torch.empty((N, Ho, Wo, Co), device=x_raw.device, dtype=torch.float32)
    chunks = L.query("edrl_conv_stats_chunks", N, Ho, Wo)
    part = torch.empty((chunks, 3, Co), device=x_raw.device, dtype=torch.float32)
    _launch_timed("conv_gather", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_fwd_bnin_stats_f32", P(x_raw),
                  P(in_fcoef), P(w), P(out), P(part), part.numel() * 4, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad,
                  nbytes=4.0 * (x_raw.numel() + w.numel() + out.numel()))
    return out, part, chunks


def conv2d_dgrad_bn(g, yraw, bcoef, wt, x_shape, stride=1, pad=0, out=None, accumulate=False, ep=None):
    """dx (+)= conv_transpose(d_raw(g, yraw; bcoef)).  ep = (raw, mask_bytes | None, fcoef, relu) of the BatchNorm whose output
    dx is the gradient of: dx is then masked in the epilogue and the partial sums of that BatchNorm's backward are returned.
    -> dx  |  (dx, part [chunks][2][Ci], chunks)."""
    N, Hi, Wi, Ci = x_shape
    _, Ho, Wo, Co = g.shape
    KH, KW = wt.shape[1], wt.shape[2]
    if out is None:
        out = torch.empty((N, Hi, Wi, Ci), device=g.device, dtype=torch.float32)
        accumulate = False
    part, chunks, nbytes = None, 0, 0
    if ep is not None:
        chunks = L.query("edrl_conv_dgrad_bn_chunks", N, Hi, Wi, stride, pad)
        part = torch.empty((chunks, 2, Ci), device=g.device, dtype=torch.float32)
        nbytes = part.numel() * 4
    ep_raw, ep_mask, ep_fcoef, ep_relu = ep if ep is not None else (None, None, None, False)
    kernels = _dgrad_kernels(Hi, Wi, KH, KW, stride, pad, accumulate and ep is None)
    _launch_timed("conv_gather", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_dgrad_bn_f32", P(g), P(yraw), P(bcoef),
                  P(wt), P(out), N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, FLAG_ACCUM if accumulate else 0, P(ep_raw),
                  P(ep_mask), P(ep_fcoef), 1 if ep_relu else 0, P(part), nbytes, kernels=kernels,
                  nbytes=4.0 * ((2 if yraw is not None else 1) * g.numel() + wt.numel() + out.numel() * (2 if accumulate else 1) +
                                (out.numel() if ep is not None else 0)))
    return out if ep is None else (out, part, chunks)


def conv2d_wgrad_bn(g, yraw, bcoef, x, x_fcoef, w_shape, stride=1, pad=0):
    """dw = sum_pixels d_raw(g, yraw; bcoef) (x) X with X = relu(bn(x; x_fcoef)) (x_fcoef None: x as is)."""
    N, Hi, Wi, Ci = x.shape
    _, Ho, Wo, Co = g.shape
    KH, KW = w_shape[1], w_shape[2]
    out = torch.empty(w_shape, device=g.device, dtype=torch.float32)
    nbytes = L.query("edrl_conv2d_nhwc_wgrad_workspace_bytes", N, Ho, Wo, Co, Ci, KH, KW)
    ws = torch.empty(nbytes // 4, device=g.device, dtype=torch.float32)
    _launch_timed("conv_wgrad", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_wgrad_bn_f32", P(g), P(yraw), P(bcoef),
                  P(x), P(x_fcoef), P(out), P(ws), nbytes, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, 0,
                  nbytes=4.0 * ((2 if yraw is not None else 1) * g.numel() + x.numel() + out.numel()))
    return out


def linear_fwd(x2, w, bias=None, mul=None, relu=False):
    """x2 [rows, in] (row stride allowed), w [out, in] -> [rows, out]."""
    rows, cin = x2.shape
    cout = w.shape[0]
    out = torch.empty((rows, cout), device=x2.device, dtype=torch.float32)
    _launch_timed("linear_gather", 2.0 * rows * cin * cout, "edrl_conv2d_nhwc_fwd_f32", P(x2), P(w), P(bias), P(mul),
                  P(out), rows, 1, 1, cin, 1, 1, cout, 1, 1, 1, 0, _ld(x2), cout, cout, FLAG_RELU if relu else 0)
    return out


# Permuted [in, out] copies of the head's Linear weights for the input gradients: inside a train_step (train.train_step brackets
# it with step_cache_begin / _end) every weight is permuted once and the copy serves both views' backward passes; outside, per call.
_perm_cache = None


def step_cache_begin():
    global _perm_cache
    _perm_cache = {}


def step_cache_end():
    global _perm_cache
    _perm_cache = None


def _permuted(w):
    if _perm_cache is None:
        return permute_weight(w)
    key = (w.data_ptr(), tuple(w.shape), torch.cuda.current_stream().cuda_stream)
    t = _perm_cache.get(key)
    if t is None:
        t = _perm_cache[key] = permute_weight(w)
    return t


def linear_dgrad(dy2, w):
    """dy2 [rows, out], w [out, in] -> dx [rows, in]."""
    rows, cout = dy2.shape
    cin = w.shape[1]
    wt = _permuted(w)  # [in, out]
    out = torch.empty((rows, cin), device=dy2.device, dtype=torch.float32)
    _launch_timed("linear_gather", 2.0 * rows * cin * cout, "edrl_conv2d_nhwc_dgrad_f32", P(dy2), P(wt), P(out), rows,
                  1, 1, cin, 1, 1, cout, 1, 1, 1, 0, _ld(dy2), cin, 0)
    return out


def matmul_tn(a2, b2, out=None):
    """a2 [rows, M], b2 [rows, N] (row strides allowed) -> a2^T @ b2  [M, N] (split-K over rows); `out`: contiguous [M, N] target."""
    rows, M = a2.shape
    N = b2.shape[1]
    if out is None:
        out = torch.empty((M, N), device=a2.device, dtype=torch.float32)
    nbytes = L.query("edrl_conv2d_nhwc_wgrad_workspace_bytes", rows, 1, 1, M, N, 1, 1)
    ws = torch.empty(max(nbytes // 4, 1), device=a2.device, dtype=torch.float32)
    _launch_timed("linear_wgrad", 2.0 * rows * M * N, "edrl_conv2d_nhwc_wgrad_f32", P(a2), P(b2), P(out), P(ws), nbytes,
                  rows, 1, 1, N, 1, 1, M, 1, 1, 1, 0, _ld(a2), _ld(b2), 0)
    return out


def sum_axis1(x3, scale=1.0, out=None):
    A, Ln, D = x3.shape
    if out is None:
        out = torch.empty((A, D), device=x3.device, dtype=torch.float32)
    L.call("edrl_sum_axis1_f32", P(x3), P(out), A, Ln, D, float(scale))
    return out


def bcast_axis1(x2, Ln, scale=1.0):
    A, D = x2.shape
    out = torch.empty((A, Ln, D), device=x2.device, dtype=torch.float32)
    L.call("edrl_bcast_axis1_f32", P(x2), P(out), A, Ln, D, float(scale), 0)
    return out


def colsum(x2, out=None):
    """[rows, D] -> [D]; `out`: contiguous [D] target."""
    x2 = x2.contiguous()
    return sum_axis1(x2.view(1, x2.shape[0], x2.shape[1]), out=None if out is None else out.view(1, -1)).view(-1)


# ------------------------------------------------------------------ autograd functions
class LinearFn(torch.autograd.Function):
    """y = (relu?)(x @ w^T + b) * mask   — nn.Linear (+ReLU +Dropout mask), fusion_net.py:82-90 etc."""

    @staticmethod
    def forward(ctx, x, w, b, relu, mask):
        _chk(x, "linear.x", contiguous=False); _chk(w, "linear.w")
        x2 = _rows2d(x)
        m2 = None if mask is None else _chk(mask, "linear.mask").reshape(-1, w.shape[0])
        y = linear_fwd(x2, w, b, m2, relu)
        ctx.relu = relu
        ctx.has_bias = b is not None
        ctx.save_for_backward(x2, w, y if relu else None, m2)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w, y, m2 = ctx.saved_tensors
        g = _rows2d(dy.contiguous())
        if ctx.relu:
            g = ew(EW_MASKED_BWD, g, m2, y)
        elif m2 is not None:
            g = ew(EW_MUL, g, m2)
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(g, w).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dw = matmul_tn(g, x2)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(g)
        return dx, dw, db, None, None


def linear(x, w, b=None, relu=False, mask=None):
    return LinearFn.apply(x, w, b, relu, mask)


class InProjFn(torch.autograd.Function):
    """The packed in-projection of nn.MultiheadAttention for key is value (fusion_net.py:569-570, 733-743):
    q = x @ W[:E]^T + b[:E],  kv = y @ W[E:]^T + b[E:]  with W = in_proj_weight [3E, E].  One node, so that the packed weight and
    bias get ONE gradient tensor written slice by slice; slicing the parameter in Python instead makes autograd build a
    zero-filled [3E, E] tensor per slice, copy the slice gradient in and add the two (5 launches and 3 x 12.6 MB of traffic per
    parameter and call: 80 launches per step for the four attention blocks and two views)."""

    @staticmethod
    def forward(ctx, x, y, w, b, E):
        _chk(x, "in_proj.x", contiguous=False); _chk(y, "in_proj.y", contiguous=False); _chk(w, "in_proj.w"); _chk(b, "in_proj.b")
        x2, y2 = _rows2d(x), _rows2d(y)
        q = linear_fwd(x2, w[:E], b[:E])
        kv = linear_fwd(y2, w[E:], b[E:])
        ctx.save_for_backward(x2, y2, w)
        ctx.E, ctx.xshape, ctx.yshape = E, x.shape, y.shape
        return q.view(*x.shape[:-1], E), kv.view(*y.shape[:-1], 2 * E)

    @staticmethod
    def backward(ctx, dq, dkv):
        x2, y2, w = ctx.saved_tensors
        E = ctx.E
        gq, gkv = _rows2d(dq.contiguous()), _rows2d(dkv.contiguous())
        dx = dy = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = linear_dgrad(gq, w[:E]).view(ctx.xshape)
        if ctx.needs_input_grad[1]:
            dy = linear_dgrad(gkv, w[E:]).view(ctx.yshape)
        if ctx.needs_input_grad[2]:
            dw = torch.empty_like(w)
            matmul_tn(gq, x2, out=dw[:E])
            matmul_tn(gkv, y2, out=dw[E:])
        if ctx.needs_input_grad[3]:
            db = torch.empty((3 * E,), device=w.device, dtype=torch.float32)
            colsum(gq, out=db[:E])
            colsum(gkv, out=db[E:])
        return dx, dy, dw, db, None


def in_proj(x, y, w, b, E):
    return InProjFn.apply(x, y, w, b, E)


class ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = ew(EW_RELU, _chk(x, "relu.x", False))
        ctx.save_for_backward(y)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return ew(EW_RELU_BWD, dy.contiguous(), y).view(dy.shape)


def relu(x):
    return ReluFn.apply(x)


class AddFn(torch.autograd.Function):
    """a + b, optionally followed by ReLU (AttentionModel residuals, fusion_net.py:572,575-576)."""

    @staticmethod
    def forward(ctx, a, b, do_relu):
        y = ew(EW_ADD_RELU if do_relu else EW_AXPBY, _chk(a, "add.a", False), _chk(b, "add.b", False).contiguous())
        ctx.do_relu = do_relu
        if do_relu:
            ctx.save_for_backward(y)
        return y.view(a.shape)

    @staticmethod
    def backward(ctx, dy):
        if ctx.do_relu:
            (y,) = ctx.saved_tensors
            g = ew(EW_RELU_BWD, dy.contiguous(), y).view(dy.shape)
            return g, g, None
        return dy, dy, None


def add(a, b, relu=False):
    return AddFn.apply(a, b, relu)


class SoftplusFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "softplus.x", False).contiguous()
        ctx.save_for_backward(x)
        return ew(EW_SOFTPLUS, x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ew(EW_SOFTPLUS_BWD, dy.contiguous(), x)


def softplus(x):
    return SoftplusFn.apply(x)


class L2NormAxis1Fn(torch.autograd.Function):
    """F.normalize(x, dim=1) for x [A, L, D] (fusion_net.py:149-150)."""

    @staticmethod
    def forward(ctx, x, eps):
        x = _chk(x, "l2norm.x", False).contiguous()
        A, Ln, D = x.shape
        y = torch.empty_like(x)
        inv = torch.empty((A, D), device=x.device, dtype=torch.float32)
        L.call("edrl_l2norm_axis1_fwd_f32", P(x), P(y), P(inv), A, Ln, D, float(eps))
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        A, Ln, D = y.shape
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        L.call("edrl_l2norm_axis1_bwd_f32", P(dy), P(y), P(inv), P(dx), A, Ln, D)
        return dx, None


def l2norm_axis1(x, eps=1e-12):
    return L2NormAxis1Fn.apply(x, eps)


class MeanAxis1Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _chk(x, "mean.x", False).contiguous()
        ctx.Ln = x.shape[1]
        return sum_axis1(x, 1.0 / x.shape[1])

    @staticmethod
    def backward(ctx, dy):
        return bcast_axis1(dy.contiguous(), ctx.Ln, 1.0 / ctx.Ln)


def mean_axis1(x):
    return MeanAxis1Fn.apply(x)


class RepeatAxis1Fn(torch.autograd.Function):
    """[A, D] -> [A, L, D] (mu_proxy.repeat(B,1,1) is the A=1 case, fusion_net.py:246-247)."""

    @staticmethod
    def forward(ctx, x, Ln):
        return bcast_axis1(_chk(x, "repeat.x", False).contiguous(), Ln)

    @staticmethod
    def backward(ctx, dy):
        return sum_axis1(dy.contiguous()), None


def repeat_axis1(x, Ln):
    return RepeatAxis1Fn.apply(x, Ln)


class AffineBcastFn(torch.autograd.Function):
    """out[a,l,d] = u[a,d] + v[a,d] * w[a,l,d]  (w = noise, no gradient)."""

    @staticmethod
    def forward(ctx, u, v, w):
        u = _chk(u, "affine.u", False).contiguous(); v = _chk(v, "affine.v", False).contiguous()
        w = _chk(w, "affine.w", False).contiguous()
        A, Ln, D = w.shape
        out = torch.empty_like(w)
        L.call("edrl_affine_bcast_fwd_f32", P(u), P(v), P(w), P(out), A, Ln, D)
        ctx.save_for_backward(w)
        return out

    @staticmethod
    def backward(ctx, dout):
        (w,) = ctx.saved_tensors
        A, Ln, D = w.shape
        dout = dout.contiguous()
        du = torch.empty((A, D), device=w.device, dtype=torch.float32)
        dv = torch.empty_like(du)
        L.call("edrl_affine_bcast_bwd_f32", P(dout), P(w), P(du), P(dv), A, Ln, D)
        return du, dv, None


def affine_bcast(u, v, w):
    return AffineBcastFn.apply(u, v, w)


class TopkMarginFn(torch.autograd.Function):
    """EPRL proxy loss from attention scores (fusion_net.py:227-243). att [B,C,S], y int64 [B]."""

    @staticmethod
    def forward(ctx, att, y, K):
        att = _chk(att, "topk.att", False).contiguous()
        B, C, S = att.shape
        sel = torch.zeros((B, C, S), device=att.device, dtype=torch.uint8)
        means = torch.empty((B, 2), device=att.device, dtype=torch.float32)
        e = torch.empty((B,), device=att.device, dtype=torch.float32)
        loss = torch.empty((1,), device=att.device, dtype=torch.float32)
        L.call("edrl_topk_margin_fwd_f32", P(att), P(y), P(sel), P(means), P(e), P(loss), B, C, S, K)
        ctx.save_for_backward(sel, e, y)
        ctx.dims = (B, C, S, K)
        ctx.mark_non_differentiable(sel)
        return loss.view(()), sel

    @staticmethod
    def backward(ctx, dloss, _dsel):
        sel, e, y = ctx.saved_tensors
        B, C, S, K = ctx.dims
        datt = torch.empty((B, C, S), device=e.device, dtype=torch.float32)
        dloss = dloss.contiguous().view(1)
        L.call("edrl_topk_margin_bwd_f32", P(dloss), P(e), P(sel), P(y), P(datt), B, C, S, K)
        return datt, None, None


def topk_margin(att, y, K=100):
    return TopkMarginFn.apply(att, y, K)


class Poe2Fn(torch.autograd.Function):
    """PoE.forward for two modalities (fusion_net.py:26-52): returns mu + var."""

    @staticmethod
    def forward(ctx, mu0, s0, mu1, s1, phi, eps):
        ts = [_chk(t, "poe", False).contiguous() for t in (mu0, s0, mu1, s1)]
        phi = _chk(phi, "poe.phi")
        out = torch.empty_like(ts[0])
        L.call("edrl_poe2_fwd_f32", P(ts[0]), P(ts[1]), P(ts[2]), P(ts[3]), P(phi), P(out), out.numel(), float(eps))
        ctx.save_for_backward(*ts, phi)
        ctx.eps = eps
        return out

    @staticmethod
    def backward(ctx, g):
        mu0, s0, mu1, s1, phi = ctx.saved_tensors
        g = g.contiguous()
        outs = [torch.empty_like(mu0) for _ in range(4)]
        dphi = torch.empty_like(phi)
        ws = torch.empty(128, device=g.device, dtype=torch.float32)
        L.call("edrl_poe2_bwd_f32", P(g), P(mu0), P(s0), P(mu1), P(s1), P(phi), P(outs[0]), P(outs[1]), P(outs[2]),
               P(outs[3]), P(dphi), P(ws), g.numel(), float(ctx.eps))
        return outs[0], outs[1], outs[2], outs[3], dphi, None


def poe2(mu0, s0, mu1, s1, phi, eps=1e-8):
    return Poe2Fn.apply(mu0, s0, mu1, s1, phi, eps)


class KlNormalFn(torch.autograd.Function):
    """MedFusion.get_KL_loss (fusion_net.py:838-850, 390-402). mu, sg [B, C, D] -> scalar."""

    @staticmethod
    def forward(ctx, mu, sg):
        mu = _chk(mu, "kl.mu", False).contiguous(); sg = _chk(sg, "kl.sg", False).contiguous()
        Bn, C, D = mu.shape
        loss = torch.empty((1,), device=mu.device, dtype=torch.float32)
        ws = torch.empty(64, device=mu.device, dtype=torch.float32)
        L.call("edrl_kl_normal_fwd_f32", P(mu), P(sg), P(loss), P(ws), Bn, C, D)
        ctx.save_for_backward(mu, sg)
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        mu, sg = ctx.saved_tensors
        Bn, C, D = mu.shape
        dmu = torch.empty_like(mu); dsg = torch.empty_like(sg)
        dloss = dloss.contiguous().view(1)
        L.call("edrl_kl_normal_bwd_f32", P(dloss), P(mu), P(sg), P(dmu), P(dsg), Bn, C, D)
        return dmu, dsg


def kl_normal(mu, sg):
    return KlNormalFn.apply(mu, sg)


class MhaCoreFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh)) v per head. q [B,Lq,E], kv [B,N,2E] -> ctx [B,Lq,E]."""

    @staticmethod
    def forward(ctx, q, kv, H):
        q = _chk(q, "mha.q", False).contiguous(); kv = _chk(kv, "mha.kv", False).contiguous()
        B, Lq, E = q.shape
        N = kv.shape[1]
        Pm = torch.empty((B, H, Lq, N), device=q.device, dtype=torch.float32)
        out = torch.empty_like(q)
        L.call("edrl_mha_core_fwd_f32", P(q), P(kv), P(Pm), P(out), B, Lq, N, H, E)
        ctx.save_for_backward(q, kv, Pm)
        ctx.H = H
        return out

    @staticmethod
    def backward(ctx, dctx):
        q, kv, Pm = ctx.saved_tensors
        B, Lq, E = q.shape
        N = kv.shape[1]
        dctx = dctx.contiguous()
        dq = torch.empty_like(q); dkv = torch.empty_like(kv)
        L.call("edrl_mha_core_bwd_f32", P(dctx), P(q), P(kv), P(Pm), P(dq), P(dkv), B, Lq, N, ctx.H, E)
        return dq, dkv, None


def mha_core(q, kv, H):
    return MhaCoreFn.apply(q, kv, H)


class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, eps):
        x = _chk(x, "ln.x", False).contiguous()
        E = x.shape[-1]
        R = x.numel() // E
        y = torch.empty_like(x)
        mean = torch.empty((R,), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        L.call("edrl_layernorm_fwd_f32", P(x), P(w), P(b), P(y), P(mean), P(rstd), R, E, float(eps))
        ctx.save_for_backward(x, w, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, mean, rstd = ctx.saved_tensors
        E = x.shape[-1]
        R = x.numel() // E
        dy = dy.contiguous()
        dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(w)
        L.call("edrl_layernorm_bwd_f32", P(dy), P(x), P(w), P(mean), P(rstd), P(dx), P(dw), P(db), R, E)
        return dx, dw, db, None


def layernorm(x, w, b, eps=1e-5):
    return LayerNormFn.apply(x, w, b, eps)


def _bn_ws(M, C, device, extra=0):
    nbytes = L.query("edrl_bn_workspace_bytes", M, C) + extra
    return torch.empty(nbytes // 4, device=device, dtype=torch.float32), nbytes


class BatchNorm1dTrainFn(torch.autograd.Function):
    """Train-mode BatchNorm1d(affine=False) on [M, C]; `updates` running-stat updates per call
    (DILR applies bn1/bn2 twice per forward to the same tensor: fusion_net.py:658,757-758)."""

    @staticmethod
    def forward(ctx, x, running_mean, running_var, momentum, eps, updates):
        x = _chk(x, "bn.x", False).contiguous()
        M, C = x.shape
        mean = torch.empty((C,), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean); scale = torch.empty_like(mean); shift = torch.empty_like(mean)
        ws, nbytes = _bn_ws(M, C, x.device)
        for _ in range(updates):
            L.call("edrl_bn_train_stats_f32", P(x), M, C, C, None, None, P(running_mean), P(running_var),
                   float(momentum), float(eps), P(mean), P(rstd), P(scale), P(shift), P(ws), nbytes)
        y = torch.empty_like(x)
        L.call("edrl_bn_apply_f32", P(x), P(mean), P(scale), P(shift), None, P(y), None, M, C, C, 0)
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        M, C = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        ws, nbytes = _bn_ws(M, C, x.device, extra=2 * C * 4)
        L.call("edrl_bn_bwd_f32", P(dy), None, None, P(x), P(mean), P(rstd), None, None, None, 0, P(dx), None, 0, M, C, C,
               P(ws), nbytes)
        return dx, None, None, None, None, None


def batchnorm1d_train(x, running_mean, running_var, momentum=0.1, eps=1e-5, updates=1):
    return BatchNorm1dTrainFn.apply(x, running_mean, running_var, momentum, eps, updates)


class CrossCorrFn(torch.autograd.Function):
    """c = scale * a^T @ b for column blocks a, b [rows, n] (row strides allowed)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        a2 = _rows2d(_chk(a, "xcorr.a", False)); b2 = _rows2d(_chk(b, "xcorr.b", False))
        c = matmul_tn(a2, b2)
        if scale != 1.0:
            c = ew(EW_SCALE, c, alpha=scale)
        ctx.save_for_backward(a2, b2)
        ctx.scale = scale
        return c

    @staticmethod
    def backward(ctx, dc):
        a2, b2 = ctx.saved_tensors
        dcs = ew(EW_SCALE, dc.contiguous(), alpha=ctx.scale)
        # da[r][i] = sum_j dc[i][j] b[r][j] ;  db[r][j] = sum_i dc[i][j] a[r][i]
        da = linear_fwd(b2, dcs)
        db = linear_fwd(a2, permute_weight(dcs))
        return da, db, None


def cross_corr(a, b, scale):
    return CrossCorrFn.apply(a, b, scale)


class BtLossFn(torch.autograd.Function):
    """(loss_c + loss_u)/2 of DILR.bt_loss_cross given the two diagonal blocks of c (fusion_net.py:664-677,754)."""

    @staticmethod
    def forward(ctx, cc, cu, lambd):
        cc = _chk(cc, "bt.cc"); cu = _chk(cu, "bt.cu")
        n = cc.shape[0]
        out = torch.empty(7, device=cc.device, dtype=torch.float32)
        ws = torch.empty(512, device=cc.device, dtype=torch.float32)
        L.call("edrl_bt_loss_fwd_f32", P(cc), P(cu), n, float(lambd), P(out), P(ws))
        ctx.save_for_backward(cc, cu)
        ctx.lambd = lambd
        ctx.mark_non_differentiable(out)
        return out[6].clone(), out

    @staticmethod
    def backward(ctx, dloss, _dparts):
        cc, cu = ctx.saved_tensors
        n = cc.shape[0]
        dcc = torch.empty_like(cc); dcu = torch.empty_like(cu)
        dloss = dloss.contiguous().view(1)
        L.call("edrl_bt_loss_bwd_f32", P(dloss), P(cc), P(cu), P(dcc), P(dcu), n, float(ctx.lambd))
        return dcc, dcu, None


def bt_loss(cc, cu, lambd=0.0051):
    return BtLossFn.apply(cc, cu, lambd)


class SmoothCeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, y, smoothing):
        pred = _chk(pred, "ce.pred", False).contiguous()
        B, C = pred.shape
        loss = torch.empty((1,), device=pred.device, dtype=torch.float32)
        L.call("edrl_smooth_ce_fwd_f32", P(pred), P(y), P(loss), B, C, float(smoothing))
        ctx.save_for_backward(pred, y)
        ctx.smoothing = smoothing
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        pred, y = ctx.saved_tensors
        B, C = pred.shape
        dpred = torch.empty_like(pred)
        dloss = dloss.contiguous().view(1)
        L.call("edrl_smooth_ce_bwd_f32", P(dloss), P(pred), P(y), P(dpred), B, C, float(ctx.smoothing))
        return dpred, None, None


def smooth_ce(pred, y, smoothing=0.1):
    return SmoothCeFn.apply(pred, y, smoothing)


def argmax_rows(x):
    x = _chk(x.detach(), "argmax.x", False).contiguous()
    out = torch.empty((x.shape[0],), device=x.device, dtype=torch.int64)
    L.call("edrl_argmax_rows_f32", P(x), P(out), x.shape[0], x.shape[1])
    return out


class ScalarMixFn(torch.autograd.Function):
    """sum_i w_i * s_i over 0-d tensors (loss mixers, fusion_net.py:870-879; fusion_train.py:212)."""

    @staticmethod
    def forward(ctx, weights, *scalars):
        import ctypes
        n = len(scalars)
        ss = [_chk(s, "mix.s", False).contiguous() for s in scalars]
        ptrs = (ctypes.c_void_p * n)(*[s.data_ptr() for s in ss])
        ws = (ctypes.c_float * n)(*[float(w) for w in weights])
        out = torch.empty((1,), device=ss[0].device, dtype=torch.float32)
        L.call("edrl_scalar_mix_f32", ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(ws, ctypes.c_void_p), n, P(out))
        ctx.weights = [float(w) for w in weights]
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous().view(1)
        return (None,) + tuple(ew(EW_SCALE, g, alpha=w).view(()) for w in ctx.weights)


def scalar_mix(weights, scalars):
    return ScalarMixFn.apply(tuple(weights), *scalars)


class MkMmdFn(torch.autograd.Function):
    """MK_MMD(source, target, kernel_mul, kernel_num)  (code/MMD.py:46-74)."""

    @staticmethod
    def forward(ctx, source, target, kernel_mul, kernel_num):
        _chk(source, "mmd.source", False); _chk(target, "mmd.target", False)
        total = torch.cat([source, target], dim=0).contiguous()
        n, d = total.shape
        ns = source.shape[0]
        G = linear_fwd(total, total)
        sq = torch.empty((n,), device=total.device, dtype=torch.float32)
        L.call("edrl_rowsq_f32", P(total), P(sq), n, d, d)
        loss = torch.empty((1,), device=total.device, dtype=torch.float32)
        saved = torch.empty((3,), device=total.device, dtype=torch.float32)
        L.call("edrl_mk_mmd_fwd_f32", P(G), P(sq), n, ns, float(kernel_mul), int(kernel_num), P(loss), P(saved))
        ctx.save_for_backward(total, G, sq, saved)
        ctx.cfg = (ns, float(kernel_mul), int(kernel_num))
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        total, G, sq, saved = ctx.saved_tensors
        ns, mul, num = ctx.cfg
        n, d = total.shape
        dloss = dloss.contiguous().view(1)
        ws = torch.empty((n, n), device=total.device, dtype=torch.float32)
        coef = torch.empty((n, n), device=total.device, dtype=torch.float32)
        L.call("edrl_mk_mmd_bwd_f32", P(dloss), P(G), P(sq), P(saved), n, ns, mul, num, P(ws), P(coef))
        dtotal = linear_fwd(coef, permute_weight(total))  # coef [n,n] @ total [n,d]
        return dtotal[:ns], dtotal[ns:], None, None


def mk_mmd(source, target, kernel_mul=2.0, kernel_num=5):
    return MkMmdFn.apply(source, target, kernel_mul, kernel_num)


# ------------------------------------------------------------------ forward-only helpers of the eval path (no autograd)
def softmax_rows(x2):
    x2 = _chk(x2.detach(), "softmax.x", False).contiguous()
    y = torch.empty_like(x2)
    L.call("edrl_softmax_rows_f32", P(x2), P(y), x2.shape[0], x2.shape[1])
    return y


def rowmean(x2):
    """[R, D] -> [R] mean over the last axis."""
    x2 = _chk(x2.detach(), "rowmean.x", False).contiguous()
    out = torch.empty((x2.shape[0],), device=x2.device, dtype=torch.float32)
    L.call("edrl_rowsum_f32", P(x2), P(out), x2.shape[0], x2.shape[1], x2.shape[1], 1.0 / x2.shape[1])
    return out


def pseudo_label(combined, threshold):
    combined = _chk(combined.detach(), "pseudo.x", False).contiguous()
    B, C = combined.shape
    labels = torch.empty((B,), device=combined.device, dtype=torch.int64)
    keep = torch.empty((B,), device=combined.device, dtype=torch.uint8)
    count = torch.zeros((1,), device=combined.device, dtype=torch.int32)
    L.call("edrl_pseudo_label_f32", P(combined), B, C, float(threshold), P(labels), P(keep), P(count))
    return labels, keep, count


def entropy_rows(x2):
    x2 = _chk(x2.detach(), "entropy.x", False).contiguous()
    out = torch.empty((1,), device=x2.device, dtype=torch.float32)
    L.call("edrl_entropy_rows_f32", P(x2), P(out), x2.shape[0], x2.shape[1])
    return out.view(())


def batchnorm_eval(x, running_mean, running_var, gamma=None, beta=None, eps=1e-5, relu=False, residual=None):
    """Eval-mode BatchNorm (running statistics) on [..., C] rows, optional residual add and ReLU."""
    x = _chk(x.detach(), "bn_eval.x", False).contiguous()
    C = x.shape[-1]
    M = x.numel() // C
    scale = torch.empty((C,), device=x.device, dtype=torch.float32)
    shift = torch.empty_like(scale)
    L.call("edrl_bn_eval_params_f32", P(gamma), P(beta), P(running_var), float(eps), P(scale), P(shift), C)
    y = torch.empty_like(x)
    L.call("edrl_bn_apply_f32", P(x), P(running_mean), P(scale), P(shift), P(residual), P(y), None, M, C, C,
           1 if relu else 0)
    return y


class KlRowsFn(torch.autograd.Function):
    """compute_kl_divergence(p, m) (code/MMD.py:92-95)."""

    @staticmethod
    def forward(ctx, p, m):
        p = _chk(p, "kl.p", False).contiguous(); m = _chk(m, "kl.m", False).contiguous()
        out = torch.empty((1,), device=p.device, dtype=torch.float32)
        L.call("edrl_kl_rows_fwd_f32", P(p), P(m), P(out), p.shape[0], p.shape[1])
        ctx.save_for_backward(p, m)
        return out.view(())

    @staticmethod
    def backward(ctx, dloss):
        p, m = ctx.saved_tensors
        dp = torch.empty_like(p); dm = torch.empty_like(m)
        L.call("edrl_kl_rows_bwd_f32", P(dloss.contiguous().view(1)), P(p), P(m), P(dp), P(dm), p.shape[0], p.shape[1])
        return dp, dm


class AxpbyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, alpha, beta):
        ctx.ab = (alpha, beta)
        return ew(EW_AXPBY, _chk(a, "axpby.a", False), _chk(b, "axpby.b", False).contiguous(), alpha=alpha, beta=beta).view(a.shape)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        return ew(EW_SCALE, g, alpha=ctx.ab[0]).view(g.shape), ew(EW_SCALE, g, alpha=ctx.ab[1]).view(g.shape), None, None


def twin_view(x, sigma=0.5, noise=None):
    """Device-side high-noise view: clip(x + sigma*N(0,1), 0, 1) (data_harvard.py:769-783); RNG draw by torch."""
    x = _chk(x, "twin.x", False).contiguous()
    if noise is None:
        noise = torch.randn_like(x)
    out = torch.empty_like(x)
    L.call("edrl_twin_view_f32", P(x), P(noise), P(out), x.numel(), float(sigma))
    return out


def salt_pepper_(x, amount, coords=None, generator=None):
    """In-place salt-and-pepper noise on an NCHW batch [N,C,H,W] (OCT volumes: pass the [B*S,1,H,W] slice view), the
    device form of add_salt_peper / add_salt_peper_3D (data_harvard.py:24-48): ceil(amount*H*W*0.5) salt points set
    to 1 on every channel, then as many pepper points set to 0, coordinates uniform in [0, H-1) x [0, W-1) (numpy
    randint's exclusive upper bound i-1, reproduced).  `coords` = (salt_rows, salt_cols, pepper_rows, pepper_cols)
    int32 [N, n] lets the caller supply the draws (parity tests); otherwise they are drawn with torch on the device."""
    x = _chk(x, "salt_pepper.x", False)
    if not x.is_contiguous() or x.dim() != 4:
        raise RuntimeError("salt_pepper_: contiguous [N,C,H,W] expected")
    N, C, H, W = x.shape
    n = int(-(-amount * H * W * 0.5 // 1))
    if coords is None:
        draw = lambda hi: torch.randint(0, max(hi - 1, 1), (N, n), device=x.device, dtype=torch.int32, generator=generator)
        coords = (draw(H), draw(W), draw(H), draw(W))
    sr, sc, pr, pc = [c.to(device=x.device, dtype=torch.int32).contiguous() for c in coords]
    L.call("edrl_scatter_fill_nchw_f32", P(x), P(sr), P(sc), N, sr.shape[1], C, H, W, 1.0)
    L.call("edrl_scatter_fill_nchw_f32", P(x), P(pr), P(pc), N, pr.shape[1], C, H, W, 0.0)
    return x


# ------------------------------------------------------------------ bf16 contractions (C2/C4 precision; raw helpers)
def to_bf16(x):
    x = _chk(x, "to_bf16.x", False).contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    L.call("edrl_cast_f32_to_bf16", P(x), P(out), x.numel())
    return out


def to_f32(x):
    x = x.contiguous()
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    L.call("edrl_cast_bf16_to_f32", P(x), P(out), x.numel())
    return out


def permute_weight_bf16(w):
    """fp32 [Co,KH,KW,Ci] -> bf16 [Ci,KH,KW,Co]."""
    A, B, C = w.shape[0], w.shape[1] * w.shape[2], w.shape[3]
    out = torch.empty((C, w.shape[1], w.shape[2], A), device=w.device, dtype=torch.bfloat16)
    L.call("edrl_permute_weight_bf16", P(w), P(out), A, B, C)
    return out


def conv2d_fwd_bf16(x, w, stride=1, pad=0, stats=False):
    """x bf16 [N,Hi,Wi,Ci], w bf16 [Co,KH,KW,Ci] -> y bf16 (, part fp32 [chunks][3][Co], chunks)."""
    N, Hi, Wi, Ci = x.shape
    Co, KH, KW, _ = w.shape
    Ho = (Hi + 2 * pad - KH) // stride + 1
    Wo = (Wi + 2 * pad - KW) // stride + 1
    out = torch.empty((N, Ho, Wo, Co), device=x.device, dtype=torch.bfloat16)
    part, chunks = None, 0
    if stats:
        chunks = L.query("edrl_conv_stats_chunks", N, Ho, Wo)
        part = torch.empty((chunks, 3, Co), device=x.device, dtype=torch.float32)
    # algorithmic bytes: a strided 1x1 layer needs only the pixels it samples
    x_read = x.numel() / (stride * stride) if (KH == 1 and KW == 1) else x.numel()
    _launch_timed("conv_gather_bf16", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_fwd_bf16", P(x), P(w), P(out),
                  P(part), part.numel() * 4 if stats else 0, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad,
                  nbytes=2.0 * (x_read + w.numel() + out.numel()))
    return (out, part, chunks) if stats else out


def conv2d_dgrad_bf16(dy, wt, x_shape, stride=1, pad=0, out=None, accumulate=False):
    N, Hi, Wi, Ci = x_shape
    _, Ho, Wo, Co = dy.shape
    KH, KW = wt.shape[1], wt.shape[2]
    if out is None:
        out = torch.empty((N, Hi, Wi, Ci), device=dy.device, dtype=torch.bfloat16)
    _launch_timed("conv_gather_bf16", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_dgrad_bf16", P(dy), P(wt),
                  P(out), N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, FLAG_ACCUM if accumulate else 0,
                  kernels=_dgrad_kernels(Hi, Wi, KH, KW, stride, pad, accumulate),
                  nbytes=2.0 * (dy.numel() + wt.numel() + out.numel() * (2 if accumulate else 1)))
    return out


def conv2d_wgrad_bf16(dy, x, w_shape, stride=1, pad=0, out=None, accumulate=False):
    """dy bf16 [N,Ho,Wo,Co], x bf16 [N,Hi,Wi,Ci] -> dw fp32 [Co,KH,KW,Ci]."""
    N, Hi, Wi, Ci = x.shape
    _, Ho, Wo, Co = dy.shape
    KH, KW = w_shape[1], w_shape[2]
    if out is None:
        out = torch.empty(w_shape, device=dy.device, dtype=torch.float32)
        accumulate = False
    nbytes = L.query("edrl_conv2d_nhwc_wgrad_bf16_workspace_bytes", N, Ho, Wo, Co, Ci, KH, KW)
    ws = torch.empty(nbytes // 4, device=dy.device, dtype=torch.float32)
    x_read = x.numel() / (stride * stride) if (KH == 1 and KW == 1) else x.numel()
    _launch_timed("conv_wgrad_bf16", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_wgrad_bf16", P(dy), P(x),
                  P(out), P(ws), nbytes, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, 1 if accumulate else 0,
                  nbytes=2.0 * (dy.numel() + x_read) + 4.0 * out.numel())
    return out


# ---- fused-BatchNorm variants of the bf16 trunk (bf16 tensors, fp32 coefficient arrays; see conv2d_*_bn above)
def conv_fused_ok_bf16(N, Hi, Wi, Ci, Co, KH, stride, pad):
    Ho = (Hi + 2 * pad - KH) // stride + 1
    Wo = (Wi + 2 * pad - KH) // stride + 1
    return bool(L.query("edrl_conv2d_fused_ok_bf16", N, Hi, Wi, Ci, Ho, Wo, Co, KH, KH, stride, pad))


def conv2d_fwd_bnin_stats_bf16(x_raw, in_fcoef, w, stride=1, pad=0):
    N, Hi, Wi, Ci = x_raw.shape
    Co, KH, KW, _ = w.shape
    Ho = (Hi + 2 * pad - KH) // stride + 1
    Wo = (Wi + 2 * pad - KW) // stride + 1
    out = torch.empty((N, Ho, Wo, Co), device=x_raw.device, dtype=torch.bfloat16)
    chunks = L.query("edrl_conv_stats_chunks", N, Ho, Wo)
    part = torch.empty((chunks, 3, Co), device=x_raw.device, dtype=torch.float32)
    _launch_timed("conv_gather_bf16", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_fwd_bnin_stats_bf16", P(x_raw),
                  P(in_fcoef), P(w), P(out), P(part), part.numel() * 4, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad,
                  nbytes=2.0 * (x_raw.numel() + w.numel() + out.numel()))
    return out, part, chunks


def conv2d_dgrad_bn_bf16(g, yraw, bcoef, wt, x_shape, stride=1, pad=0, out=None, accumulate=False, ep=None):
    N, Hi, Wi, Ci = x_shape
    _, Ho, Wo, Co = g.shape
    KH, KW = wt.shape[1], wt.shape[2]
    if out is None:
        out = torch.empty((N, Hi, Wi, Ci), device=g.device, dtype=torch.bfloat16)
        accumulate = False
    part, chunks, nbytes = None, 0, 0
    if ep is not None:
        chunks = L.query("edrl_conv_dgrad_bn_chunks", N, Hi, Wi, stride, pad)
        part = torch.empty((chunks, 2, Ci), device=g.device, dtype=torch.float32)
        nbytes = part.numel() * 4
    ep_raw, ep_mask, ep_fcoef, ep_relu = ep if ep is not None else (None, None, None, False)
    kernels = _dgrad_kernels(Hi, Wi, KH, KW, stride, pad, accumulate and ep is None)
    _launch_timed("conv_gather_bf16", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_dgrad_bn_bf16", P(g), P(yraw),
                  P(bcoef), P(wt), P(out), N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, FLAG_ACCUM if accumulate else 0,
                  P(ep_raw), P(ep_mask), P(ep_fcoef), 1 if ep_relu else 0, P(part), nbytes, kernels=kernels,
                  nbytes=2.0 * ((2 if yraw is not None else 1) * g.numel() + wt.numel() + out.numel() * (2 if accumulate else 1) +
                                (out.numel() if ep is not None else 0)))
    return out if ep is None else (out, part, chunks)


def conv1x1_k64_bwd_ok_bf16(N, H, W, Ci, Co):
    return bool(L.query("edrl_conv1x1_k64_bwd_ok_bf16", N, H, W, Ci, Co))


def conv1x1_k64_bwd_bf16(g, yraw, bcoef, x2raw, x2_fcoef, wt):
    """Both gradients of an expanding 1x1 layer (64 -> 256) inside a fused-BatchNorm block from one pass over (g, yraw)
    (csrc/conv1x1_bwd_bf16.hip).  -> (dw fp32 [Co,1,1,Ci], g2 bf16 [N,H,W,Ci] masked with bn(x2raw)'s ReLU decision,
    part [chunks][2][Ci] = (sum g2, sum g2*(x2raw - mean)), chunks)."""
    N, H, W, Co = g.shape
    Ci = x2raw.shape[-1]
    dev = g.device
    g2 = torch.empty((N, H, W, Ci), device=dev, dtype=torch.bfloat16)
    chunks = L.query("edrl_conv1x1_k64_bwd_chunks", N, H, W)
    part = torch.empty((chunks, 2, Ci), device=dev, dtype=torch.float32)
    nbytes = L.query("edrl_conv1x1_k64_bwd_workspace_bytes", N, H, W)
    ws = torch.empty(nbytes // 4, device=dev, dtype=torch.float32)
    dw = torch.empty((Co, 1, 1, Ci), device=dev, dtype=torch.float32)
    M = N * H * W
    _launch_timed("conv_bwd_k64_bf16", 4.0 * M * Co * Ci, "edrl_conv1x1_k64_bwd_bf16", P(g), P(yraw), P(bcoef), P(x2raw), P(x2_fcoef),
                  P(wt), P(g2), P(part), part.numel() * 4, P(dw), P(ws), nbytes, N, H, W, Ci, Co, kernels=2,
                  nbytes=2.0 * (2 * g.numel() + x2raw.numel() + g2.numel() + wt.numel()) + 4.0 * dw.numel())
    return dw, g2, part, chunks


def conv2d_wgrad_bn_bf16(g, yraw, bcoef, x, x_fcoef, w_shape, stride=1, pad=0):
    N, Hi, Wi, Ci = x.shape
    _, Ho, Wo, Co = g.shape
    KH, KW = w_shape[1], w_shape[2]
    out = torch.empty(w_shape, device=g.device, dtype=torch.float32)
    nbytes = L.query("edrl_conv2d_nhwc_wgrad_bf16_workspace_bytes", N, Ho, Wo, Co, Ci, KH, KW)
    ws = torch.empty(nbytes // 4, device=g.device, dtype=torch.float32)
    _launch_timed("conv_wgrad_bf16", 2.0 * N * Ho * Wo * Co * KH * KW * Ci, "edrl_conv2d_nhwc_wgrad_bn_bf16", P(g), P(yraw),
                  P(bcoef), P(x), P(x_fcoef), P(out), P(ws), nbytes, N, Hi, Wi, Ci, Ho, Wo, Co, KH, KW, stride, pad, 0,
                  nbytes=2.0 * (2 * g.numel() + x.numel()) + 4.0 * out.numel())
    return out
