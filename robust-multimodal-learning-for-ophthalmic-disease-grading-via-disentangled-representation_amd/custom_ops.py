"""`torch.ops.edrl.*`: the hot-path operators registered with torch.library (SURVEY.md §8b "What a C-ABI replacement must
export": one schema per kernel class, each forwarding to the `extern "C"` launchers of libedrl_hip.so).

The registration is done from Python (`torch.library.custom_op`, i.e. the TORCH_LIBRARY mechanism without a second compiled
extension): every op has a schema, a CUDA implementation that calls the SAME C-ABI launchers the `ops.py` autograd Functions
call, a fake (meta) implementation for shape inference / tracing, and -- for the differentiable ones -- an autograd formula
composed of the backward launchers.  No CPU implementation is registered: calling an op on CPU tensors raises (the hot path
is HIP-only).

    edrl::conv2d_nhwc(x, w, stride, pad) -> y                         fusion_net.py:884-885 (encoder convs), autograd
    edrl::conv2d_nhwc_dgrad(dy, wt, x_shape, stride, pad) -> dx
    edrl::conv2d_nhwc_wgrad(dy, x, w_shape, stride, pad) -> dw
    edrl::gemm_bias_act(x, w, bias?, mask?, relu) -> y                nn.Linear (+ReLU +Dropout mask) fusion_net.py:82-90, autograd
    edrl::bn_relu_fwd(x, gamma, beta, running_mean, running_var, momentum, eps, relu, residual?) -> (y, mean, rstd, mask)
    edrl::bn_relu_bwd(dy, mask?, x, mean, rstd, gamma, want_dres) -> (dx, dgamma, dbeta, dres)
    edrl::mk_mmd(source, target, kernel_mul, kernel_num) -> loss      code/MMD.py:46-74, autograd
    edrl::smooth_ce(pred, y, smoothing) -> loss                       fusion_net.py:931-939, autograd
    edrl::eprl_proxy_loss(att, y, k) -> (loss, sel)                   fusion_net.py:227-243, autograd
    edrl::bt_cross_loss(cc, cu, lambd) -> (loss, parts)               fusion_net.py:664-677, autograd
"""
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib as L
from . import ops

P = L.ptr
_DEV = "cuda"


def _conv_out_hw(Hi, Wi, KH, KW, stride, pad):
    return (Hi + 2 * pad - KH) // stride + 1, (Wi + 2 * pad - KW) // stride + 1


# ---------------------------------------------------------------------------------------------- convolution
@torch.library.custom_op("edrl::conv2d_nhwc", mutates_args=(), device_types=_DEV)
def conv2d_nhwc(x: Tensor, w: Tensor, stride: int, pad: int) -> Tensor:
    return ops.conv2d_fwd(ops._chk(x, "conv2d_nhwc.x"), ops._chk(w, "conv2d_nhwc.w"), stride=stride, pad=pad)


@conv2d_nhwc.register_fake
def _(x, w, stride, pad):
    Ho, Wo = _conv_out_hw(x.shape[1], x.shape[2], w.shape[1], w.shape[2], stride, pad)
    return x.new_empty((x.shape[0], Ho, Wo, w.shape[0]))


@torch.library.custom_op("edrl::conv2d_nhwc_dgrad", mutates_args=(), device_types=_DEV)
def conv2d_nhwc_dgrad(dy: Tensor, wt: Tensor, x_shape: List[int], stride: int, pad: int) -> Tensor:
    return ops.conv2d_dgrad(ops._chk(dy, "dgrad.dy"), ops._chk(wt, "dgrad.wt"), tuple(x_shape), stride, pad)


@conv2d_nhwc_dgrad.register_fake
def _(dy, wt, x_shape, stride, pad):
    return dy.new_empty(tuple(x_shape))


@torch.library.custom_op("edrl::conv2d_nhwc_wgrad", mutates_args=(), device_types=_DEV)
def conv2d_nhwc_wgrad(dy: Tensor, x: Tensor, w_shape: List[int], stride: int, pad: int) -> Tensor:
    return ops.conv2d_wgrad(ops._chk(dy, "wgrad.dy"), ops._chk(x, "wgrad.x"), tuple(w_shape), stride, pad)


@conv2d_nhwc_wgrad.register_fake
def _(dy, x, w_shape, stride, pad):
    return dy.new_empty(tuple(w_shape))


def _conv_setup(ctx, inputs, output):
    x, w, stride, pad = inputs
    ctx.save_for_backward(x, w)
    ctx.geo = (stride, pad)


def _conv_backward(ctx, dy):
    x, w = ctx.saved_tensors
    stride, pad = ctx.geo
    dy = dy.contiguous()
    dx = dw = None
    if ctx.needs_input_grad[0]:
        dx = torch.ops.edrl.conv2d_nhwc_dgrad(dy, ops.permute_weight(w), list(x.shape), stride, pad)
    if ctx.needs_input_grad[1]:
        dw = torch.ops.edrl.conv2d_nhwc_wgrad(dy, x, list(w.shape), stride, pad)
    return dx, dw, None, None


conv2d_nhwc.register_autograd(_conv_backward, setup_context=_conv_setup)


# ---------------------------------------------------------------------------------------------- Linear (+ReLU +mask)
@torch.library.custom_op("edrl::gemm_bias_act", mutates_args=(), device_types=_DEV)
def gemm_bias_act(x: Tensor, w: Tensor, bias: Optional[Tensor], mask: Optional[Tensor], relu: bool) -> Tensor:
    x2 = ops._rows2d(ops._chk(x, "gemm.x", contiguous=False))
    m2 = None if mask is None else ops._chk(mask, "gemm.mask").reshape(-1, w.shape[0])
    return ops.linear_fwd(x2, ops._chk(w, "gemm.w"), bias, m2, relu).view(*x.shape[:-1], w.shape[0])


@gemm_bias_act.register_fake
def _(x, w, bias, mask, relu):
    return x.new_empty((*x.shape[:-1], w.shape[0]))


def _gemm_setup(ctx, inputs, output):
    x, w, bias, mask, relu = inputs
    ctx.save_for_backward(x, w, output if relu else None, mask)
    ctx.relu, ctx.has_bias = relu, bias is not None


def _gemm_backward(ctx, dy):
    x, w, y, mask = ctx.saved_tensors
    g = ops._rows2d(dy.contiguous())
    if ctx.relu:
        g = ops.ew(ops.EW_MASKED_BWD, g, None if mask is None else mask.reshape(-1, w.shape[0]), y.reshape(-1, w.shape[0]))
    elif mask is not None:
        g = ops.ew(ops.EW_MUL, g, mask.reshape(-1, w.shape[0]))
    x2 = ops._rows2d(x)
    dx = ops.linear_dgrad(g, w).view(x.shape) if ctx.needs_input_grad[0] else None
    dw = ops.matmul_tn(g, x2) if ctx.needs_input_grad[1] else None
    db = ops.colsum(g) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
    return dx, dw, db, None, None


gemm_bias_act.register_autograd(_gemm_backward, setup_context=_gemm_setup)


# ---------------------------------------------------------------------------------------------- BatchNorm (+residual +ReLU)
@torch.library.custom_op("edrl::bn_relu_fwd", mutates_args=("running_mean", "running_var"), device_types=_DEV)
def bn_relu_fwd(x: Tensor, gamma: Tensor, beta: Tensor, running_mean: Tensor, running_var: Tensor, momentum: float, eps: float,
                relu: bool, residual: Optional[Tensor]) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    from .encoders import _bn_fwd
    bn = {"weight": gamma, "bias": beta, "running_mean": running_mean, "running_var": running_var, "momentum": momentum, "eps": eps}
    out, mean, rstd, mask = _bn_fwd(ops._chk(x, "bn.x"), bn, relu, residual)
    if mask is None:
        mask = torch.empty((0,), device=x.device, dtype=torch.uint8)
    return out, mean, rstd, mask


@bn_relu_fwd.register_fake
def _(x, gamma, beta, running_mean, running_var, momentum, eps, relu, residual):
    C = x.shape[-1]
    M = x.numel() // C
    return (torch.empty_like(x), x.new_empty((C,)), x.new_empty((C,)),
            x.new_empty((M, C // 4) if relu else (0,), dtype=torch.uint8))


@torch.library.custom_op("edrl::bn_relu_bwd", mutates_args=(), device_types=_DEV)
def bn_relu_bwd(dy: Tensor, mask: Optional[Tensor], x: Tensor, mean: Tensor, rstd: Tensor, gamma: Tensor,
                want_dres: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    from .encoders import _bn_bwd
    d_raw, dg, db, dres = _bn_bwd(ops._chk(dy, "bn_bwd.dy"), mask, x, mean, rstd, gamma, want_dres)
    if dres is None:
        dres = torch.empty((0,), device=x.device, dtype=torch.float32)
    return d_raw, dg, db, dres


@bn_relu_bwd.register_fake
def _(dy, mask, x, mean, rstd, gamma, want_dres):
    C = x.shape[-1]
    return torch.empty_like(x), x.new_empty((C,)), x.new_empty((C,)), (torch.empty_like(x) if want_dres else x.new_empty((0,)))


# ---------------------------------------------------------------------------------------------- losses
@torch.library.custom_op("edrl::mk_mmd_fwd", mutates_args=(), device_types=_DEV)
def mk_mmd_fwd(source: Tensor, target: Tensor, kernel_mul: float, kernel_num: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    ops._chk(source, "mmd.source", False); ops._chk(target, "mmd.target", False)
    total = torch.cat([source, target], dim=0).contiguous()
    n, d = total.shape
    G = ops.linear_fwd(total, total)
    sq = torch.empty((n,), device=total.device, dtype=torch.float32)
    L.call("edrl_rowsq_f32", P(total), P(sq), n, d, d)
    loss = torch.empty((), device=total.device, dtype=torch.float32)
    saved = torch.empty((3,), device=total.device, dtype=torch.float32)
    L.call("edrl_mk_mmd_fwd_f32", P(G), P(sq), n, source.shape[0], float(kernel_mul), int(kernel_num), P(loss), P(saved))
    return loss, total, G, sq, saved


@mk_mmd_fwd.register_fake
def _(source, target, kernel_mul, kernel_num):
    n = source.shape[0] + target.shape[0]
    return (source.new_empty(()), source.new_empty((n, source.shape[1])), source.new_empty((n, n)), source.new_empty((n,)),
            source.new_empty((3,)))


@torch.library.custom_op("edrl::mk_mmd_bwd", mutates_args=(), device_types=_DEV)
def mk_mmd_bwd(dloss: Tensor, total: Tensor, G: Tensor, sq: Tensor, saved: Tensor, ns: int, kernel_mul: float,
               kernel_num: int) -> Tensor:
    n = total.shape[0]
    ws = torch.empty((n, n), device=total.device, dtype=torch.float32)
    coef = torch.empty((n, n), device=total.device, dtype=torch.float32)
    L.call("edrl_mk_mmd_bwd_f32", P(dloss.contiguous().view(1)), P(G), P(sq), P(saved), n, ns, float(kernel_mul), int(kernel_num),
           P(ws), P(coef))
    return ops.linear_fwd(coef, ops.permute_weight(total))


@mk_mmd_bwd.register_fake
def _(dloss, total, G, sq, saved, ns, kernel_mul, kernel_num):
    return torch.empty_like(total)


def _mmd_setup(ctx, inputs, output):
    source, target, kernel_mul, kernel_num = inputs
    _, total, G, sq, saved = output
    ctx.save_for_backward(total, G, sq, saved)
    ctx.cfg = (source.shape[0], kernel_mul, kernel_num)


def _mmd_backward(ctx, dloss, *_unused):
    total, G, sq, saved = ctx.saved_tensors
    ns, mul, num = ctx.cfg
    dtotal = torch.ops.edrl.mk_mmd_bwd(dloss, total, G, sq, saved, ns, mul, num)
    return dtotal[:ns], dtotal[ns:], None, None


mk_mmd_fwd.register_autograd(_mmd_backward, setup_context=_mmd_setup)


def mk_mmd(source, target, kernel_mul=2.0, kernel_num=5):
    """MK_MMD through the registered ops (same launchers as ops.mk_mmd)."""
    return torch.ops.edrl.mk_mmd_fwd(source, target, kernel_mul, kernel_num)[0]


@torch.library.custom_op("edrl::smooth_ce", mutates_args=(), device_types=_DEV)
def smooth_ce(pred: Tensor, y: Tensor, smoothing: float) -> Tensor:
    pred = ops._chk(pred, "ce.pred", False).contiguous()
    loss = torch.empty((), device=pred.device, dtype=torch.float32)
    L.call("edrl_smooth_ce_fwd_f32", P(pred), P(y), P(loss), pred.shape[0], pred.shape[1], float(smoothing))
    return loss


@smooth_ce.register_fake
def _(pred, y, smoothing):
    return pred.new_empty(())


@torch.library.custom_op("edrl::smooth_ce_bwd", mutates_args=(), device_types=_DEV)
def smooth_ce_bwd(dloss: Tensor, pred: Tensor, y: Tensor, smoothing: float) -> Tensor:
    pred = pred.contiguous()
    dpred = torch.empty_like(pred)
    L.call("edrl_smooth_ce_bwd_f32", P(dloss.contiguous().view(1)), P(pred), P(y), P(dpred), pred.shape[0], pred.shape[1],
           float(smoothing))
    return dpred


@smooth_ce_bwd.register_fake
def _(dloss, pred, y, smoothing):
    return torch.empty_like(pred)


def _ce_setup(ctx, inputs, output):
    pred, y, smoothing = inputs
    ctx.save_for_backward(pred, y)
    ctx.smoothing = smoothing


def _ce_backward(ctx, dloss):
    pred, y = ctx.saved_tensors
    return torch.ops.edrl.smooth_ce_bwd(dloss, pred, y, ctx.smoothing), None, None


smooth_ce.register_autograd(_ce_backward, setup_context=_ce_setup)


@torch.library.custom_op("edrl::eprl_proxy_loss", mutates_args=(), device_types=_DEV)
def eprl_proxy_loss(att: Tensor, y: Tensor, k: int) -> Tuple[Tensor, Tensor, Tensor]:
    att = ops._chk(att, "topk.att", False).contiguous()
    B, C, S = att.shape
    sel = torch.zeros((B, C, S), device=att.device, dtype=torch.uint8)
    means = torch.empty((B, 2), device=att.device, dtype=torch.float32)
    e = torch.empty((B,), device=att.device, dtype=torch.float32)
    loss = torch.empty((), device=att.device, dtype=torch.float32)
    L.call("edrl_topk_margin_fwd_f32", P(att), P(y), P(sel), P(means), P(e), P(loss), B, C, S, k)
    return loss, sel, e


@eprl_proxy_loss.register_fake
def _(att, y, k):
    return att.new_empty(()), att.new_empty(att.shape, dtype=torch.uint8), att.new_empty((att.shape[0],))


@torch.library.custom_op("edrl::eprl_proxy_loss_bwd", mutates_args=(), device_types=_DEV)
def eprl_proxy_loss_bwd(dloss: Tensor, e: Tensor, sel: Tensor, y: Tensor, k: int) -> Tensor:
    B, C, S = sel.shape
    datt = torch.empty((B, C, S), device=e.device, dtype=torch.float32)
    L.call("edrl_topk_margin_bwd_f32", P(dloss.contiguous().view(1)), P(e), P(sel), P(y), P(datt), B, C, S, k)
    return datt


@eprl_proxy_loss_bwd.register_fake
def _(dloss, e, sel, y, k):
    return e.new_empty(sel.shape)


def _pl_setup(ctx, inputs, output):
    att, y, k = inputs
    _, sel, e = output
    ctx.save_for_backward(sel, e, y)
    ctx.k = k


def _pl_backward(ctx, dloss, *_unused):
    sel, e, y = ctx.saved_tensors
    return torch.ops.edrl.eprl_proxy_loss_bwd(dloss, e, sel, y, ctx.k), None, None


eprl_proxy_loss.register_autograd(_pl_backward, setup_context=_pl_setup)


@torch.library.custom_op("edrl::bt_cross_loss", mutates_args=(), device_types=_DEV)
def bt_cross_loss(cc: Tensor, cu: Tensor, lambd: float) -> Tuple[Tensor, Tensor]:
    cc = ops._chk(cc, "bt.cc"); cu = ops._chk(cu, "bt.cu")
    out = torch.empty(7, device=cc.device, dtype=torch.float32)
    ws = torch.empty(512, device=cc.device, dtype=torch.float32)
    L.call("edrl_bt_loss_fwd_f32", P(cc), P(cu), cc.shape[0], float(lambd), P(out), P(ws))
    return out[6].clone(), out


@bt_cross_loss.register_fake
def _(cc, cu, lambd):
    return cc.new_empty(()), cc.new_empty((7,))


@torch.library.custom_op("edrl::bt_cross_loss_bwd", mutates_args=(), device_types=_DEV)
def bt_cross_loss_bwd(dloss: Tensor, cc: Tensor, cu: Tensor, lambd: float) -> Tuple[Tensor, Tensor]:
    dcc = torch.empty_like(cc); dcu = torch.empty_like(cu)
    L.call("edrl_bt_loss_bwd_f32", P(dloss.contiguous().view(1)), P(cc), P(cu), P(dcc), P(dcu), cc.shape[0], float(lambd))
    return dcc, dcu


@bt_cross_loss_bwd.register_fake
def _(dloss, cc, cu, lambd):
    return torch.empty_like(cc), torch.empty_like(cu)


def _bt_setup(ctx, inputs, output):
    cc, cu, lambd = inputs
    ctx.save_for_backward(cc, cu)
    ctx.lambd = lambd


def _bt_backward(ctx, dloss, *_unused):
    cc, cu = ctx.saved_tensors
    dcc, dcu = torch.ops.edrl.bt_cross_loss_bwd(dloss, cc, cu, ctx.lambd)
    return dcc, dcu, None


bt_cross_loss.register_autograd(_bt_backward, setup_context=_bt_setup)

REGISTERED = ["conv2d_nhwc", "conv2d_nhwc_dgrad", "conv2d_nhwc_wgrad", "gemm_bias_act", "bn_relu_fwd", "bn_relu_bwd", "mk_mmd_fwd",
              "mk_mmd_bwd", "smooth_ce", "smooth_ce_bwd", "eprl_proxy_loss", "eprl_proxy_loss_bwd", "bt_cross_loss", "bt_cross_loss_bwd"]
