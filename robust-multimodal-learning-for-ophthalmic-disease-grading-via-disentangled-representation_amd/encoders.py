"""Build-owned CNN image encoders for the two EDRL encoder slots.

The reference's encoders (`Models.fundus_swin_network.build_model`, `Models.unetr.UNETR_base_3DNet`,
fusion_net.py:1-2,796,799) are absent from the reference repository; it only fixes the contract
    transformer_2DNet(X[0]) -> (tokens [B, N2, 1024], pooled)
    transformer_3DNet(X[1]) -> (tokens [B, N3,  768], pooled)          (fusion_net.py:884-885)
BASELINE.json names ResNet-18 / ResNet-50: fundus 2D, OCT as a slice stack (B*S single-channel
images through a 2D ResNet, one token per slice).  Topology: He et al. v1.5 (7x7/2 stem, 3x3/2
max-pool, [2,2,2,2] basic or [3,4,6,3] bottleneck blocks with the stride on the 3x3).

MI355X layout: activations NHWC fp32, conv weights [Co,KH,KW,Ci]; the whole trunk is ONE
autograd node whose forward/backward sequence the HIP launchers directly (implicit-GEMM MFMA
convs, two-level BN reductions, fused BN+ReLU(+residual) apply) — no per-layer autograd graph.
"""
import math
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import ops

P = L.ptr

_CFG = {
    18: ("basic", [2, 2, 2, 2], 1),
    34: ("basic", [3, 4, 6, 3], 1),
    50: ("bottleneck", [3, 4, 6, 3], 4),
}


# EDRL_WGRAD_STREAM=1 issues the weight-gradient kernels on a side stream (measured +2 % images/s at C1: they overlap
# the HBM-bound BatchNorm-backward kernels).  Off by default: concurrent streams inflate every per-kernel duration
# (HIP events and rocprof alike), which would blur the roofline measurement of the dominant kernel.
_WGRAD_SIDE_STREAM = os.environ.get("EDRL_WGRAD_STREAM", "0") == "1"   # wgrad beside the next BatchNorm backward


def _bn_ws(M, C, device, extra=0):
    nbytes = L.query("edrl_bn_workspace_bytes", M, C) + extra
    return torch.empty(nbytes // 4, device=device, dtype=torch.float32), nbytes


def _bn_fwd(raw, bn, relu, residual=None):
    """raw [N,H,W,C] -> (out, mean, rstd). Train-mode statistics; running stats updated in place."""
    C = raw.shape[-1]
    M = raw.numel() // C
    dev = raw.device
    mean = torch.empty(C, device=dev, dtype=torch.float32)
    rstd = torch.empty_like(mean); scale = torch.empty_like(mean); shift = torch.empty_like(mean)
    ws, nbytes = _bn_ws(M, C, dev)
    L.call("edrl_bn_train_stats_f32", P(raw), M, C, C, P(bn["weight"]), P(bn["bias"]), P(bn["running_mean"]),
           P(bn["running_var"]), float(bn["momentum"]), float(bn["eps"]), P(mean), P(rstd), P(scale), P(shift),
           P(ws), nbytes)
    out = torch.empty_like(raw)
    mask = torch.empty((M, C // 4), device=dev, dtype=torch.uint8) if relu else None
    L.call("edrl_bn_apply_f32", P(raw), P(mean), P(scale), P(shift), P(residual), P(out), P(mask), M, C, C,
           1 if relu else 0)
    return out, mean, rstd, mask


_FUSE_STATS = os.environ.get("EDRL_FUSE_BN_STATS", "1") != "0"
_STEM_S2D = os.environ.get("EDRL_STEM_S2D", "1") != "0"


def _conv_bn_fwd(inp, w, bn, stride, pad, relu, residual=None):
    """conv -> train-mode BN (-> +residual) (-> ReLU).  When the vector fast path applies (Ci % 16 == 0) the BN
    statistics come out of the conv epilogue (no extra pass over the conv output).  -> (raw, out, mean, rstd, mask)."""
    if not (_FUSE_STATS and inp.shape[-1] % 16 == 0 and w.shape[0] % 4 == 0):
        raw = ops.conv2d_fwd(inp, w, stride=stride, pad=pad)
        out, mean, rstd, mask = _bn_fwd(raw, bn, relu, residual)
        return raw, out, mean, rstd, mask
    raw, part, chunks = ops.conv2d_fwd_stats(inp, w, bn["running_mean"], stride, pad)
    C = raw.shape[-1]
    M = raw.numel() // C
    dev = raw.device
    mean = torch.empty(C, device=dev, dtype=torch.float32)
    rstd = torch.empty_like(mean); scale = torch.empty_like(mean); shift = torch.empty_like(mean)
    gbytes = L.query("edrl_bn_finalize_group_ws_bytes", chunks, C)
    gws = torch.empty(gbytes // 8, device=dev, dtype=torch.float64)
    L.call("edrl_bn_finalize_partials_f32", P(part), chunks, 128, M, C, P(bn["weight"]), P(bn["bias"]),
           P(bn["running_mean"]), P(bn["running_var"]), float(bn["momentum"]), float(bn["eps"]), P(mean), P(rstd),
           P(scale), P(shift), P(gws), gbytes)
    out = torch.empty_like(raw)
    mask = torch.empty((M, C // 4), device=dev, dtype=torch.uint8) if relu else None
    L.call("edrl_bn_apply_f32", P(raw), P(mean), P(scale), P(shift), P(residual), P(out), P(mask), M, C, C,
           1 if relu else 0)
    return raw, out, mean, rstd, mask


def _bn_bwd(dout, mask, raw, mean, rstd, gamma, want_dres):
    """-> (d_raw, dgamma, dbeta, dres). `mask` = the ReLU sign-bit bytes written by _bn_fwd (None: no ReLU)."""
    C = raw.shape[-1]
    M = raw.numel() // C
    dev = raw.device
    d_raw = torch.empty_like(raw)
    dgamma = torch.empty(C, device=dev, dtype=torch.float32)
    dbeta = torch.empty_like(dgamma)
    dres = torch.empty_like(raw) if want_dres else None
    ws, nbytes = _bn_ws(M, C, dev, extra=2 * C * 4)
    L.call("edrl_bn_bwd_f32", P(dout), None, P(mask), P(raw), P(mean), P(rstd), P(gamma), P(dgamma), P(dbeta), 0,
           P(d_raw), P(dres), 0, M, C, C, P(ws), nbytes)
    return d_raw, dgamma, dbeta, dres


# ---------------------------------------------------------------------------------------------------------------------
# Fused-BatchNorm blocks (EDRL_FUSE_BN=1, default).  Inside a residual block the BatchNorm passes ride on the convolutions:
#   forward   conv_k+1 reads the RAW output of conv_k and forms relu(bn_k(.)) in its operand load (ops.conv2d_fwd_bnin_stats);
#             only the block output relu(bn_last(c_last) + identity) is materialised (edrl_bn_apply_res_f32, which also
#             normalises the raw downsample branch in place of a stored copy);
#   backward  each data-gradient kernel masks its result with the ReLU decision of the BatchNorm below (recomputed from the
#             raw tensor, or the sign bytes of the block output) and emits that BatchNorm's partial sums from its epilogue;
#             d_raw = A*g - K1 - K2*(x-mean) is formed in the operand loads of the next dgrad / wgrad kernels.
# Per conv->BN unit this removes bn_apply (fwd), colstat + bn_bwd_apply (bwd), the activated copy and the d_raw tensor.
# A block whose layer geometries lack the fused fast paths (edrl_conv2d_fused_ok_f32: tiny / odd maps) keeps the separate
# passes; the gradient handed from block to block is ("plain", dout) or ("masked", g, part, chunks, planes).
_FUSE_BN = os.environ.get("EDRL_FUSE_BN", "1") != "0"
_GRAD_STASH = os.environ.get("EDRL_TRUNK_GRAD_STASH", "1") != "0"   # 0: every pass hands its parameter gradients to the autograd engine
_STEM_RAW16 = os.environ.get("EDRL_BF16_STEM_RAW16", "1") != "0"    # (bf16 trunk) raw stem conv output stored as bf16, statistics from the conv epilogue
_STEM_BF16MMA = os.environ.get("EDRL_BF16_STEM_MMA", "1") != "0"   # (bf16 trunk, 1-channel input) stem conv on the bf16 matrix pipe
_FUSE_STEM = os.environ.get("EDRL_FUSE_STEM", "1") != "0"      # BatchNorm + ReLU of the stem folded into its max-pool (fp32 trunk)


def _fcoef_from_partials(part, chunks, M, C, bn):
    """conv-epilogue chunk partials -> fcoef [5][C] = {mean, rstd, scale, shift, shift2}; running statistics updated in place."""
    dev = part.device
    fc = torch.empty((5, C), device=dev, dtype=torch.float32)
    gbytes = L.query("edrl_bn_finalize_group_ws_bytes", chunks, C)
    gws = torch.empty(max(gbytes // 8, 1), device=dev, dtype=torch.float64)
    L.call("edrl_bn_finalize_fcoef_f32", P(part), chunks, 128, M, C, P(bn["weight"]), P(bn["bias"]),
           P(bn["running_mean"]), P(bn["running_var"]), float(bn["momentum"]), float(bn["eps"]), P(fc), P(gws), gbytes)
    return fc


def _bcoef_from_partials(part, chunks, planes, M, gamma, fcoef):
    """BatchNorm-backward partial sums -> (bcoef [4][C], dgamma, dbeta)."""
    C = fcoef.shape[1]
    dev = part.device
    bc = torch.empty((4, C), device=dev, dtype=torch.float32)
    dgamma = torch.empty(C, device=dev, dtype=torch.float32)
    dbeta = torch.empty_like(dgamma)
    gbytes = L.query("edrl_bn_bwd_group_ws_bytes", chunks, C)
    gws = torch.empty(max(gbytes // 8, 1), device=dev, dtype=torch.float64)
    L.call("edrl_bn_bwd_finalize_partials_f32", P(part), chunks, planes, M, C, P(gamma), P(fcoef), P(dgamma), P(dbeta), P(bc),
           P(gws), gbytes)
    return bc, dgamma, dbeta


def _bn_bwd_reduce(K, dout, mask, raw, fcoef, want_g):
    """Standalone first half of a BatchNorm(+ReLU) backward: -> (g = dout*mask | dout itself, part, chunks, planes=3)."""
    C = raw.shape[-1]
    M = raw.numel() // C
    g = torch.empty_like(raw) if want_g else None
    ws, nbytes = _bn_ws(M, C, raw.device)
    ops.call_timed_bytes("bn_bwd_reduce", M * C * (2 * K.elt + (K.elt if want_g else 0.0) + (0.25 if mask is not None else 0.0)),
                         K.reduce_name, P(dout), P(mask), P(raw), P(fcoef), P(g), P(ws), nbytes, M, C)
    return (g if want_g else dout), ws, (M + 1023) // 1024, 3


def _dbg_act(raw, fc):
    """(tests only) the activation a fused consumer forms on the fly: relu(fma(x, scale, shift2)), ONE rounding.  Formed in fp64
    and rounded once (x*scale is exact in fp64), so that the ReLU decisions handed to the decision-pinned oracle are the
    kernels' own: torch.addcmul may round the product first, which flips the sign of a pre-activation within one ulp of zero --
    one such element moved layer1.0.bn1.bias of the ResNet-50 step test by 2e-3 (tests/test_gpu_head.py, round 4)."""
    return torch.clamp_min(_dbg_pre(raw, fc), 0.0)


def _dbg_pre(raw, fc):
    """(tests only) fma(x, scale, shift2) with the kernels' single rounding (see _dbg_act)."""
    return torch.addcmul(fc[4].double(), raw.double(), fc[2].double()).float()


def _dbg_draw(g, raw, bc):
    """(tests only) d_raw = A*g + nK2*x + C2 as the fused consumers form it."""
    return bc[0] * g + bc[1] * raw + bc[2]


# ---------------------------------------------------------------------------------------------------------------------
# Kernel sets: the trunk's forward/backward schedule (_TrunkFn) is written once; what differs between the fp32 trunk (C1/C3)
# and the bf16 trunk (C2/C4: bf16 activations / gradients and bf16 MFMA convs, fp32 BatchNorm statistics, fp32 weights and
# weight gradients, fp32 stem) is which launchers it calls.
_ZERO_C = {}


def _act_coef(K, fc):
    """(mean, shift) operands of edrl_bn_apply_mx for a materialised activation relu(bn(raw)) inside a fused block: (0, shift2), so
    that the pass forms fma(x, scale, shift2) -- the single rounding the conv kernels' operand loads and epilogues use
    (edrl_bn_pre2) -- and a unit takes the same ReLU decisions and stores the same activation, bit for bit, whether it is
    materialised (mid_sep, wide blocks) or formed in the consumer's operand load.  With (mean, shift) -- (x - mean)*scale + shift,
    rounds 1-4 -- the policies differed at ulp-level ties, which bf16 storage amplified to 5e-2 .. 1e-1 of the trunk output
    (tests/test_gpu_encoder.py::test_*_trunk_block_policies_agree: now bit-identical, fp32 and bf16)."""
    C = fc.shape[1]
    z = _ZERO_C.get((C, fc.device))
    if z is None:
        z = _ZERO_C[(C, fc.device)] = torch.zeros(C, device=fc.device, dtype=torch.float32)
    return z, fc[4]


def _f32_split_build():
    """Does the loaded library form fp32 products as bf16x3 splits (include/edrl_hip.h edrl_f32_contraction_split)?  Decides the
    default block policy of the fp32 trunk only (the fp32-MFMA build is best all-fused)."""
    return bool(L.lib().fn["edrl_f32_contraction_split"]())


class _K32:
    act_dtype = torch.float32
    conv_bn_fwd = staticmethod(lambda *a, **k: _conv_bn_fwd(*a, **k))
    bn_bwd = staticmethod(lambda *a, **k: _bn_bwd(*a, **k))
    conv_wgrad = staticmethod(lambda dy, inp, wshape, s, p: ops.conv2d_wgrad(dy, inp, wshape, s, p))
    conv_dgrad = staticmethod(lambda dy, wt, xshape, s, p, out=None, accumulate=False:
                              ops.conv2d_dgrad(dy, wt, xshape, s, p, out=out, accumulate=accumulate))
    permute = staticmethod(lambda w: ops.permute_weight(w))
    fused_ok = staticmethod(lambda *a: ops.conv_fused_ok(*a))
    fwd_stats = staticmethod(lambda inp, w, s, p: ops.conv2d_fwd_stats(inp, w, None, s, p))
    fwd_bnin_stats = staticmethod(lambda raw, fc, w, s, p: ops.conv2d_fwd_bnin_stats(raw, fc, w, s, p))
    wgrad_bn = staticmethod(lambda *a: ops.conv2d_wgrad_bn(*a))
    dgrad_bn = staticmethod(lambda *a, **k: ops.conv2d_dgrad_bn(*a, **k))
    apply_res_name, reduce_name, elt = "edrl_bn_apply_res_f32", "edrl_bn_bwd_reduce_f32", 4.0
    # Since the fp32 contractions run as bf16x3 splits on the bf16 MFMA (csrc/conv_gemm.hip EDRL_F32_SPLIT) the matrix pipe is 2.7x
    # faster per fp32 product and the operand transforms of the fused kernels are no longer hidden behind it: per layer at 1056
    # images (profiles/r05_fused_layers_split_1056img.txt) the fused form still wins on every 1x1 layer of stages 1-3 (two K-wide
    # tensors never stored) but loses on every 3x3 layer (transforms re-applied per tap: l3 3x3 4.11 -> 4.68 ms per unit) and on
    # stage 4.  So the fp32 trunk takes the bf16 trunk's policy (see _KBF16): mid_sep -- the 3x3 layer of a fused bottleneck block on
    # the plain kernels -- and above fuse_max_planes "wide" blocks (materialised activations and d_raw, BatchNorm-backward statistics
    # still from the data gradients' epilogues).  Measured on the C1 step, one box, in order (scripts/gpu_f32_policy_sweep.sh):
    # all fused 69.8 images/s (138.6 GiB) -> mid_sep 73.1 (148.8 GiB) -> + wide stage 4 73.2-73.4 (150.4) -> + wide stage 3 73.3
    # (154.5) -> + wide stage 2 72.6: mid_sep is the gain, wide blocks are neutral, so the default keeps every block fused
    # (EDRL_F32_FUSE_MAXPLANES=256 / 128: wide blocks from stage 4 / 3 up).  EDRL_F32_MID_SEP=0: the all-fused trunk of rounds 2-4
    # (what the fp32-MFMA build is best with).
    fuse_max_planes = int(os.environ.get("EDRL_F32_FUSE_MAXPLANES", str(1 << 30)))
    mid_sep = os.environ.get("EDRL_F32_MID_SEP", "1" if _f32_split_build() else "0") != "0"
    wide_sep = os.environ.get("EDRL_F32_WIDE_SEP", "1") != "0"
    mx, draw_name = 0, "edrl_bn_draw_f32"       # storage flag of the _mx entry points; the standalone d_raw pass
    # draw_sep (split build): in blocks of at least this many planes the BACKWARD of the 1x1 units takes a materialised d_raw too
    # (one edrl_bn_draw_f32 pass per unit, then the plain-operand weight / data gradient, epilogues unchanged).  The kernels that
    # form d_raw in their operand loads need 200-245 registers: 2 workgroups per CU, matrix pipe 49 % busy against 70 % of the
    # plain ones at 3 (profiles/r05_pmc_f32_split_fused_1x1.txt).  Measured on the C1 step, one box (scripts/gpu_env_sweep.sh):
    # off 78.1 images/s, from 512 planes (stage 4) 78.5, from 256 78.0, from 128 77.0 -- the extra pass over the d_raw tensors eats
    # the gain everywhere but on the 7 x 7 maps.  The forward stays fused.  EDRL_F32_DRAW_SEP_MINPLANES=1073741824: off.
    draw_sep_min_planes = int(os.environ.get("EDRL_F32_DRAW_SEP_MINPLANES", "512" if _f32_split_build() else str(1 << 30)))
    conv3_bwd_ok = staticmethod(lambda *a: False)       # (bf16 only) one-pass backward of the expanding 1x1 layers
    conv3_bwd = None
    grad_in = staticmethod(lambda dout: dout.contiguous())
    feat_out = staticmethod(lambda cur: cur)

    @staticmethod
    def stem_fwd(T, x, p, bnd, cap, cb):
        folded = False
        if _STEM_S2D and _FUSE_STEM and not x.requires_grad:
            # conv (4x4 over the space-to-depth image) -> statistics -> max-pool that applies BN+ReLU to its input on the fly:
            # the activated stem tensor (64 x 112^2 per 224^2 image) and its sign bytes are never stored
            raw, x_keep, folded = ops.stem_conv_fwd(x, p["conv1.weight"])
            bn = bnd("bn1", p)
            N, H, W, C = raw.shape
            M = N * H * W
            fc = torch.empty((5, C), device=x.device, dtype=torch.float32)
            ws, nbytes = _bn_ws(M, C, x.device)
            L.call("edrl_bn_train_stats_fcoef_f32", P(raw), M, C, P(bn["weight"]), P(bn["bias"]), P(bn["running_mean"]),
                   P(bn["running_var"]), float(bn["momentum"]), float(bn["eps"]), P(fc), P(ws), nbytes)
            Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
            p0 = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.float32)
            idx = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.uint8)
            ops.call_timed_bytes("maxpool_bn_fwd", M * C * 4.0 + p0.numel() * 5.0, "edrl_maxpool3x3s2_bn_fwd_f32", P(raw), P(fc), P(p0),
                                 P(idx), N, H, W, C)
            if cap is not None:
                a0 = _dbg_act(raw, fc)
                cap["conv1"] = dict(bn="bn1", inp=x, stride=2, pad=3, relu=True, residual=None, raw=raw, out=a0, mean=fc[0],
                                    rstd=fc[1], mask=None, fused=True)
                cap["maxpool"] = dict(inp=a0, out=p0, idx=idx, fused=True)
            return p0, ("fused", x_keep, folded, raw, fc, idx)
        if _STEM_S2D and not x.requires_grad:        # as a 4x4 conv over the space-to-depth image (ops.stem_conv_fwd)
            raw, x_keep, folded = ops.stem_conv_fwd(x, p["conv1.weight"])
            a0, m0, r0, k0 = _bn_fwd(raw, bnd("bn1", p), True)
            if cap is not None:
                cap["conv1"] = dict(bn="bn1", inp=x, stride=2, pad=3, relu=True, residual=None, raw=raw, out=a0, mean=m0,
                                    rstd=r0, mask=k0)
        else:
            raw, a0, m0, r0, k0 = cb("conv1", "bn1", x, 2, 3, True)
            x_keep = x
        N, H, W, C = a0.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        p0 = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.uint8)
        L.call("edrl_maxpool3x3s2_fwd_f32", P(a0), P(p0), P(idx), N, H, W, C)
        if cap is not None:
            cap["maxpool"] = dict(inp=a0, out=p0, idx=idx)
        return p0, (x_keep, folded, raw, a0.shape, m0, r0, k0, idx)

    @staticmethod
    def stem_bwd(T, stem, p, dcur, grads, cap, bn_bwd, conv_bwd, needs_x):
        if stem[0] == "fused":
            _, x, folded, raw, fc, idx = stem
            N, H, W, C = raw.shape
            M = N * H * W
            ws, nbytes = _bn_ws(M, C, raw.device)
            ops.call_timed_bytes("maxpool_bn_bwd", M * C * 4.0 + dcur.numel() * 5.0, "edrl_maxpool3x3s2_bn_bwd_reduce_f32", P(dcur),
                                 P(idx), P(raw), P(fc), P(ws), nbytes, N, H, W, C)
            bc, dg, db = _bcoef_from_partials(ws, (M + 1023) // 1024, 3, M, p["bn1.weight"], fc)
            grads["bn1.weight"], grads["bn1.bias"] = dg, db
            draw = torch.empty_like(raw)
            ops.call_timed_bytes("maxpool_bn_bwd", M * C * 8.0 + dcur.numel() * 5.0, "edrl_maxpool3x3s2_bn_bwd_apply_f32", P(dcur),
                                 P(idx), P(raw), P(fc), P(bc), P(draw), N, H, W, C)
            grads["conv1.weight"] = ops.stem_conv_wgrad(draw, x, tuple(p["conv1.weight"].shape), folded)
            if cap is not None:
                da0 = torch.empty_like(raw)
                L.call("edrl_maxpool3x3s2_bwd_f32", P(dcur), P(idx), P(da0), N, H, W, C)
                cap["maxpool"].update(dout=dcur, dinp=da0)
                g0 = da0 * (_dbg_pre(raw, fc) > 0)
                cap["bwd:bn1"] = dict(dout=g0, dgamma=dg, dbeta=db, d_raw=draw, dres=None, masked=True)
                cap["conv1"].update(d_raw=draw, dW=grads["conv1.weight"], dx_before=None, dx_after=None)
            return None
        x, folded, raw, a0_shape, m0, r0, k0, idx = stem
        N, H, W, C = a0_shape
        da0 = torch.empty(a0_shape, device=dcur.device, dtype=torch.float32)
        L.call("edrl_maxpool3x3s2_bwd_f32", P(dcur), P(idx), P(da0), N, H, W, C)
        if cap is not None:
            cap["maxpool"].update(dout=dcur, dinp=da0)
        draw, _ = bn_bwd("bn1", da0, k0, raw, (m0, r0))
        if folded:
            grads["conv1.weight"] = ops.stem_conv_wgrad(draw, x, tuple(p["conv1.weight"].shape), True)
            if cap is not None:
                cap["conv1"].update(d_raw=draw, dW=grads["conv1.weight"], dx_before=None, dx_after=None)
            return None
        return conv_bwd("conv1", draw, x, 2, 3, need_dx=needs_x)


# Weight shadows (round 5): inside a train_step the bf16 forward operand and the permuted data-gradient operand of every conv
# weight of a trunk are made ONCE, by one multi-tensor launch (ResNetTrunk.build_shadows <- train.train_step), instead of one
# cast + one permute launch per layer and view.  Keyed by the fp32 weight's device pointer; valid from step_begin to step_end
# (the weights do not change in between); outside a train_step the per-call casts below remain.  EDRL_WEIGHT_SHADOWS=0: off.
_WEIGHT_SHADOWS = os.environ.get("EDRL_WEIGHT_SHADOWS", "1") != "0"
_SHADOW_CAST = {}


def _wbf16(w):
    s = _SHADOW_CAST.get(w.data_ptr())
    return s if s is not None else ops.to_bf16(w)


class _KBF16:
    act_dtype = torch.bfloat16
    conv_bn_fwd = staticmethod(lambda *a, **k: _conv_bn_fwd_bf16(*a, **k))
    bn_bwd = staticmethod(lambda *a, **k: _bn_bwd_mx(*a, **k))
    conv_wgrad = staticmethod(lambda dy, inp, wshape, s, p: ops.conv2d_wgrad_bf16(dy, inp, wshape, s, p))
    conv_dgrad = staticmethod(lambda dy, wt, xshape, s, p, out=None, accumulate=False:
                              ops.conv2d_dgrad_bf16(dy, wt, xshape, s, p, out=out, accumulate=accumulate))
    permute = staticmethod(lambda w: ops.permute_weight_bf16(w))
    fused_ok = staticmethod(lambda *a: ops.conv_fused_ok_bf16(*a))
    fwd_stats = staticmethod(lambda inp, w, s, p: ops.conv2d_fwd_bf16(inp, _wbf16(w), s, p, stats=True))
    fwd_bnin_stats = staticmethod(lambda raw, fc, w, s, p: ops.conv2d_fwd_bnin_stats_bf16(raw, fc, _wbf16(w), s, p))
    wgrad_bn = staticmethod(lambda *a: ops.conv2d_wgrad_bn_bf16(*a))
    dgrad_bn = staticmethod(lambda *a, **k: ops.conv2d_dgrad_bn_bf16(*a, **k))
    apply_res_name, reduce_name, elt = "edrl_bn_apply_res_bf16", "edrl_bn_bwd_reduce_bf16", 2.0
    # Next to the bf16 MFMA (16x the fp32 rate) the operand transforms are 57-60 VALU per 8-MFMA K tile: the fused kernels turn
    # VALU-bound on the compute-heavy stages (profiles/r02_fused_layers_bf16_2112img.txt: stage-3/4 blocks lose 0.3-0.8 ms per
    # 2112 images, stage-1/2 blocks -- HBM-bound, wide BatchNorm tensors -- gain 0.7-2.5 ms), so only blocks up to this width fuse.
    fuse_max_planes = int(os.environ.get("EDRL_BF16_FUSE_MAXPLANES", "128"))
    # Inside a fused bottleneck block the 3x3 layer is the one place where fusion LOSES at bf16 (its operand transforms are
    # re-applied per tap: 26 vector instructions per MFMA, profiles/r03_pmc_traffic_c2.json; per layer sep 2.6 -> fused 3.2 ms,
    # profiles/r02_fused_layers_bf16_2112img.txt), while the 1x1 layers around it win big.  mid_sep keeps the block fused but
    # runs that one layer on the plain kernels: forward = one bn_apply pass (a1 + sign bytes) + plain conv; backward = d_raw
    # materialised by edrl_bn_draw_bf16, plain weight / data gradient, and the standalone reduce that hands the masked gradient
    # and the partial sums back to the fused chain (stride-1 blocks; the stride-2 block's reduce runs on the 4x larger map).
    mid_sep = os.environ.get("EDRL_BF16_MID_SEP", "1") != "0"
    # Bottleneck blocks ABOVE fuse_max_planes (stages 3-4: MFMA-bound layers on the 256x256 LDS-DMA cores, whose operands come by
    # DMA and cannot be transformed in registers) keep materialised activations and d_raw tensors, but their BatchNorm BACKWARD
    # statistics ride on the data gradients like in the fused blocks: every data-gradient kernel masks its result with the sign
    # bytes of the BatchNorm below and emits (sum g, sum g*(x - mean)) from its epilogue (conv_bf16_v3.hip EPI 1), d_raw =
    # A*g + nK2*x + C2 is then ONE pass (edrl_bn_draw_bf16).  Per conv -> BN unit this drops the reduction pass over (dout, raw,
    # mask) and the separate dres tensor of the block's last BatchNorm; the block output comes from bn_apply_res as in a fused block,
    # so the gradient hand-over between blocks stays ("masked", g, partials).  EDRL_BF16_WIDE_SEP=0: the separate passes of round 3.
    wide_sep = os.environ.get("EDRL_BF16_WIDE_SEP", "1") != "0"
    mx, draw_name = 1, "edrl_bn_draw_bf16"
    draw_sep_min_planes = 1 << 30
    conv3_bwd_ok = staticmethod(lambda *a: ops.conv1x1_k64_bwd_ok_bf16(*a))
    conv3_bwd = staticmethod(lambda *a: ops.conv1x1_k64_bwd_bf16(*a))
    grad_in = staticmethod(lambda dout: ops.to_bf16(dout.contiguous()))
    feat_out = staticmethod(lambda cur: ops.to_f32(cur))

    @staticmethod
    def stem_fwd(T, x, p, bnd, cap, cb):
        """fp32 stem conv + fp32 statistics, bf16 activation out.  Default (EDRL_FUSE_STEM): BatchNorm + ReLU folded into the
        max-pool as in the fp32 trunk -- the pool reads the RAW fp32 conv output and writes the bf16 pooled tensor; the activated
        112^2 tensor, its sign bytes and (in backward) its gradient never exist."""
        bn = bnd("bn1", p)
        if _FUSE_STEM and cap is None and _STEM_S2D and _STEM_RAW16:
            # the raw stem output is a bf16 tensor like every other layer's (fp32 image, fp32 MFMA, one rounding on the way out);
            # its statistics come from the conv epilogue's fp32 partials -- no separate statistics pass over the largest tensor
            if _STEM_BF16MMA and x.shape[-1] == 1 and p["conv1.weight"].shape[0] == 64 and x.shape[1] % 2 == 0 and x.shape[2] % 2 == 0:
                # 1-channel (OCT) stem on the bf16 matrix pipe: image and weights rounded to bf16 in registers (EDRL_BF16_STEM_MMA=0:
                # the fp32-MFMA stem of round 3, bf16 output only)
                raw, part, chunks, x, folded = ops.stem_conv_fwd_bf16mma(x, p["conv1.weight"])
            else:
                raw, part, chunks, x, folded = ops.stem_conv_fwd_obf16(x, p["conv1.weight"])
            N, H, W, C = raw.shape
            M = N * H * W
            fc = _fcoef_from_partials(part, chunks, M, C, bn)
            Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
            p0 = torch.empty((N, Ho, Wo, C), device=raw.device, dtype=torch.bfloat16)
            idx = torch.empty((N, Ho, Wo, C), device=raw.device, dtype=torch.uint8)
            ops.call_timed_bytes("maxpool_bn_fwd", M * C * 2.0 + p0.numel() * 3.0, "edrl_maxpool3x3s2_bn_fwd_mx", P(raw), 1, P(fc), P(p0),
                                 1, P(idx), N, H, W, C)
            return p0, ("fused", x, folded, raw, fc, idx)
        if _STEM_S2D:
            raw, x, folded = ops.stem_conv_fwd(x, p["conv1.weight"])
        else:
            raw, folded = ops.conv2d_fwd(x, p["conv1.weight"], stride=2, pad=3), False
        C = raw.shape[-1]
        M = raw.numel() // C
        dev = raw.device
        N, H, W, _ = raw.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        if _FUSE_STEM and cap is None:
            fc = torch.empty((5, C), device=dev, dtype=torch.float32)
            ws, nbytes = _bn_ws(M, C, dev)
            L.call("edrl_bn_train_stats_fcoef_f32", P(raw), M, C, P(bn["weight"]), P(bn["bias"]), P(bn["running_mean"]),
                   P(bn["running_var"]), float(bn["momentum"]), float(bn["eps"]), P(fc), P(ws), nbytes)
            p0 = torch.empty((N, Ho, Wo, C), device=dev, dtype=torch.bfloat16)
            idx = torch.empty((N, Ho, Wo, C), device=dev, dtype=torch.uint8)
            ops.call_timed_bytes("maxpool_bn_fwd", M * C * 4.0 + p0.numel() * 3.0, "edrl_maxpool3x3s2_bn_fwd_mx", P(raw), 0, P(fc), P(p0), 1,
                                 P(idx), N, H, W, C)
            return p0, ("fused", x, folded, raw, fc, idx)
        m0 = torch.empty(C, device=dev, dtype=torch.float32)
        r0 = torch.empty_like(m0); scale = torch.empty_like(m0); shift = torch.empty_like(m0)
        ws, nbytes = _bn_ws(M, C, dev)
        L.call("edrl_bn_train_stats_f32", P(raw), M, C, C, P(bn["weight"]), P(bn["bias"]), P(bn["running_mean"]),
               P(bn["running_var"]), float(bn["momentum"]), float(bn["eps"]), P(m0), P(r0), P(scale), P(shift), P(ws), nbytes)
        a0 = torch.empty(raw.shape, device=dev, dtype=torch.bfloat16)
        k0 = torch.empty((M, C // 4), device=dev, dtype=torch.uint8)
        L.call("edrl_bn_apply_mx", P(raw), 0, P(m0), P(scale), P(shift), None, P(a0), 1, P(k0), M, C, 1)
        p0 = torch.empty((N, Ho, Wo, C), device=dev, dtype=torch.bfloat16)
        idx = torch.empty((N, Ho, Wo, C), device=dev, dtype=torch.uint8)
        L.call("edrl_maxpool3x3s2_fwd_bf16", P(a0), P(p0), P(idx), N, H, W, C)
        return p0, (x, folded, raw, a0.shape, m0, r0, k0, idx)

    @staticmethod
    def stem_bwd(T, stem, p, dcur, grads, cap, bn_bwd, conv_bwd, needs_x):
        if stem[0] == "fused":
            _, x, folded, raw, fc, idx = stem
            N, H, W, C = raw.shape
            M = N * H * W
            ws, nbytes = _bn_ws(M, C, raw.device)
            r16 = 1 if raw.dtype == torch.bfloat16 else 0
            relt = 2.0 if r16 else 4.0
            ops.call_timed_bytes("maxpool_bn_bwd", M * C * relt + dcur.numel() * 3.0, "edrl_maxpool3x3s2_bn_bwd_reduce_mx", P(dcur), 1,
                                 P(idx), P(raw), r16, P(fc), P(ws), nbytes, N, H, W, C)
            bc, dg, db = _bcoef_from_partials(ws, (M + 1023) // 1024, 3, M, p["bn1.weight"], fc)
            grads["bn1.weight"], grads["bn1.bias"] = dg, db
            # d_raw in the raw tensor's storage type: the stem weight gradient (fp32 image operand) widens a bf16 d_raw on load
            draw = torch.empty(raw.shape, device=raw.device, dtype=raw.dtype)
            ops.call_timed_bytes("maxpool_bn_bwd", M * C * 2.0 * relt + dcur.numel() * 3.0, "edrl_maxpool3x3s2_bn_bwd_apply_mx", P(dcur), 1,
                                 P(idx), P(raw), r16, P(fc), P(bc), P(draw), r16, N, H, W, C)
            grads["conv1.weight"] = ops.stem_conv_wgrad(draw, x, tuple(p["conv1.weight"].shape), folded)
            return None
        x, folded, raw, a0_shape, m0, r0, k0, idx = stem
        N, H, W, C = a0_shape
        da0 = torch.empty(a0_shape, device=dcur.device, dtype=torch.bfloat16)
        L.call("edrl_maxpool3x3s2_bwd_bf16", P(dcur), P(idx), P(da0), N, H, W, C)
        draw, _ = bn_bwd("bn1", da0, k0, raw, (m0, r0))            # raw is fp32 -> fp32 gradient for the fp32 stem wgrad
        grads["conv1.weight"] = ops.stem_conv_wgrad(draw, x, tuple(p["conv1.weight"].shape), folded)
        return None


class _TrunkFn(torch.autograd.Function):
    """x NHWC fp32 [N,H,W,Cin] -> feature map NHWC fp32 [N,h,w,C].  params: flat tensor list (see ResNetTrunk).  The kernel set
    (T.kernels: _K32 | _KBF16) selects the storage type of activations / gradients and the launchers."""

    @staticmethod
    def forward(ctx, trunk, x, *params):
        T = trunk
        K = T.kernels
        p = dict(zip(T.param_names, params))
        bnd = T.bn_dict
        saved = {}
        x = ops._chk(x, "encoder input")
        cap = T._capture     # None, or a dict filled with every layer's operands/results (tests/test_gpu_layerwise.py)
        if T._capture_seq is not None:      # one fresh record per forward pass (the two views of a step: tests/test_gpu_head.py)
            cap = {}
            T._capture_seq.append(cap)
        T._wt_cache = {}     # permuted weights are shared by the backward passes that follow this forward
        ctx.step = T._step   # train_step-scoped state (ResNetTrunk.step_begin): gradient stash of the passes that share the weights
        if ctx.step is not None:
            ctx.step["nodes"] += 1

        def cb(conv_name, bn_name, inp, stride, pad, relu, residual=None):
            r = K.conv_bn_fwd(inp, p[conv_name + ".weight"], bnd(bn_name, p), stride, pad, relu, residual)
            if cap is not None:
                cap[conv_name] = dict(bn=bn_name, inp=inp, stride=stride, pad=pad, relu=relu, residual=residual, raw=r[0],
                                      out=r[1], mean=r[2], rstd=r[3], mask=r[4])
            return r

        def cf(conv_name, bn_name, inp, in_fc, stride, pad):
            """fused unit: conv over `inp` (a raw tensor + its fcoef, or an activated tensor when in_fc is None)."""
            w = p[conv_name + ".weight"]
            if in_fc is None:
                raw, part, chunks = K.fwd_stats(inp, w, stride, pad)
            else:
                raw, part, chunks = K.fwd_bnin_stats(inp, in_fc, w, stride, pad)
            C = raw.shape[-1]
            fc = _fcoef_from_partials(part, chunks, raw.numel() // C, C, bnd(bn_name, p))
            if cap is not None:
                cap[conv_name] = dict(bn=bn_name, inp=inp if in_fc is None else _dbg_act(inp, in_fc), stride=stride, pad=pad,
                                      relu=True, residual=None, raw=raw, out=_dbg_act(raw, fc), mean=fc[0], rstd=fc[1],
                                      mask=None, fused=True)
            return raw, fc

        def block_fused_ok(blk, cur):
            if not _FUSE_BN:
                return False
            N, H, W, Ci = cur.shape
            s = blk["stride"]
            pl = p[blk["name"] + ".conv1.weight"].shape[0]
            mode = "fused"
            if pl > K.fuse_max_planes:       # kernel-set policy (see _KBF16)
                if not (K.wide_sep and T.kind == "bottleneck" and cap is None):
                    return False
                mode = "wide"
            Ho, Wo = (H - 1) // s + 1, (W - 1) // s + 1
            if T.kind == "bottleneck":
                geo = [(H, W, Ci, pl, 1, 1, 0), (H, W, pl, pl, 3, s, 1), (Ho, Wo, pl, 4 * pl, 1, 1, 0)]
                co = 4 * pl
            else:
                geo = [(H, W, Ci, pl, 3, s, 1), (Ho, Wo, pl, pl, 3, 1, 1)]
                co = pl
            if blk["downsample"]:
                geo.append((H, W, Ci, co, 1, s, 0))
            return mode if all(K.fused_ok(N, h, w, ci, c_o, k, st, pd) for h, w, ci, c_o, k, st, pd in geo) else False

        def bn_act(raw, fc):
            """(wide blocks) materialised activation relu(bn(raw)) + its sign bytes, one elementwise pass."""
            C = raw.shape[-1]
            M = raw.numel() // C
            act = torch.empty_like(raw)
            kb = torch.empty((M, C // 4), device=raw.device, dtype=torch.uint8)
            mu_, sh_ = _act_coef(K, fc)
            ops.call_timed_bytes("bn_apply", M * C * (2 * K.elt + 0.25), "edrl_bn_apply_mx", P(raw), K.mx, P(mu_), P(fc[2]), P(sh_), None,
                                 P(act), K.mx, P(kb), M, C, 1)
            return act, kb

        p0, saved["stem"] = K.stem_fwd(T, x, p, bnd, cap, cb)
        cur = p0
        prev_fused = False
        for bidx, blk in enumerate(T.blocks):
            pre, s = blk["name"], blk["stride"]
            # recompute mode: the input of a block that follows a fused block is that block's output, which backward
            # rebuilds from the raw tensors (see get_x in backward) -- it is not kept
            rec = {"x": None if (T.recompute_out and prev_fused and cap is None) else cur}
            prev_fused = False
            bmode = block_fused_ok(blk, cur)
            if bmode:
                rec["fused"] = True
                prev_fused = True
                cd = fd = None
                if blk["downsample"]:
                    cd, fd = cf(pre + ".downsample.0", pre + ".downsample.1", cur, None, s, 0)
                    rec.update(cd=cd, fd=fd)
                if bmode == "wide":
                    c1, f1 = cf(pre + ".conv1", pre + ".bn1", cur, None, 1, 0)
                    a1, k1 = bn_act(c1, f1)
                    c2, f2 = cf(pre + ".conv2", pre + ".bn2", a1, None, s, 1)
                    a2, k2 = bn_act(c2, f2)
                    cl, fl = cf(pre + ".conv3", pre + ".bn3", a2, None, 1, 0)
                    rec.update(wide=True, c1=c1, f1=f1, a1=a1, k1=k1, c2=c2, f2=f2, a2=a2, k2=k2, c3=cl, f3=fl)
                    last = pre + ".conv3"
                elif T.kind == "bottleneck":
                    c1, f1 = cf(pre + ".conv1", pre + ".bn1", cur, None, 1, 0)
                    if K.mid_sep and s == 1:
                        C1 = c1.shape[-1]
                        M1 = c1.numel() // C1
                        a1 = torch.empty_like(c1)
                        k1 = torch.empty((M1, C1 // 4), device=c1.device, dtype=torch.uint8)
                        mu_, sh_ = _act_coef(K, f1)
                        ops.call_timed_bytes("bn_apply", M1 * C1 * (2 * K.elt + 0.25), "edrl_bn_apply_mx", P(c1), K.mx, P(mu_), P(f1[2]),
                                             P(sh_), None, P(a1), K.mx, P(k1), M1, C1, 1)
                        c2, f2 = cf(pre + ".conv2", pre + ".bn2", a1, None, s, 1)
                        # recompute mode: the activated copy and its sign bytes are rebuilt in backward by the same pass
                        keep_a1 = not (T.recompute_out and cap is None)
                        rec.update(a1=a1 if keep_a1 else None, k1=k1 if keep_a1 else None, mid_sep=True)
                        if cap is not None:
                            cap[pre + ".conv1"].update(out=a1, mask=k1)
                    else:
                        c2, f2 = cf(pre + ".conv2", pre + ".bn2", c1, f1, s, 1)
                    cl, fl = cf(pre + ".conv3", pre + ".bn3", c2, f2, 1, 0)
                    rec.update(c1=c1, f1=f1, c2=c2, f2=f2, c3=cl, f3=fl)
                    last = pre + ".conv3"
                else:
                    c1, f1 = cf(pre + ".conv1", pre + ".bn1", cur, None, s, 1)
                    cl, fl = cf(pre + ".conv2", pre + ".bn2", c1, f1, 1, 1)
                    rec.update(c1=c1, f1=f1, c2=cl, f2=fl)
                    last = pre + ".conv2"
                Cl = cl.shape[-1]
                Ml = cl.numel() // Cl
                out = torch.empty_like(cl)
                # recompute mode: this block's output is rebuilt in backward (by the same kernel), and so are its sign bytes
                # -- except for the last block, whose output nobody rebuilds
                keep_kl = not (T.recompute_out and cap is None and bidx + 1 < len(T.blocks))
                kl = torch.empty((Ml, Cl // 4), device=cl.device, dtype=torch.uint8) if keep_kl else None
                ops.call_timed_bytes("bn_apply_res", Ml * Cl * (3 * K.elt + (0.25 if keep_kl else 0.0)), K.apply_res_name, P(cl), P(fl),
                                     P(cd if cd is not None else cur), P(fd), P(out), P(kl), Ml, Cl, 1)   # 2 reads + 1 write + sign bytes
                rec.update(kl=kl)
                if cap is not None:
                    cap[last].update(out=out, mask=kl, residual=(cur if cd is None else
                                                                 torch.addcmul(fd[3], cd - fd[0], fd[2])))
                    if cd is not None:
                        cap[pre + ".downsample.0"].update(relu=False, out=torch.addcmul(fd[3], cd - fd[0], fd[2]))
                saved[pre] = rec
                cur = out
                continue
            if blk["downsample"]:
                cd, idn, md, rd, _ = cb(pre + ".downsample.0", pre + ".downsample.1", cur, s, 0, False)
                rec.update(cd=cd, sd=(md, rd))
            else:
                idn = cur
            if T.kind == "bottleneck":
                c1, a1, m1, r1, k1 = cb(pre + ".conv1", pre + ".bn1", cur, 1, 0, True)
                c2, a2, m2, r2, k2 = cb(pre + ".conv2", pre + ".bn2", a1, s, 1, True)
                c3, out, ml, rl, kl = cb(pre + ".conv3", pre + ".bn3", a2, 1, 0, True, residual=idn)
                rec.update(c1=c1, a1=a1, s1=(m1, r1), k1=k1, c2=c2, a2=a2, s2=(m2, r2), k2=k2, c3=c3)
            else:
                c1, a1, m1, r1, k1 = cb(pre + ".conv1", pre + ".bn1", cur, s, 1, True)
                c2, out, ml, rl, kl = cb(pre + ".conv2", pre + ".bn2", a1, 1, 1, True, residual=idn)
                rec.update(c1=c1, a1=a1, s1=(m1, r1), k1=k1, c2=c2)
            rec.update(sl=(ml, rl), kl=kl)
            saved[pre] = rec
            cur = out
        ctx.trunk = T
        ctx.saved = saved
        ctx.params = p
        ctx.needs_x = x.requires_grad
        ctx.cap = cap
        T.bump_batches_tracked()
        return K.feat_out(cur)

    @staticmethod
    def backward(ctx, dout):
        T, saved, p = ctx.trunk, ctx.saved, ctx.params
        K = T.kernels
        ctx.saved = None
        grads = {}
        cap = ctx.cap
        wt_cache = T._wt_cache

        def wt_of(name):
            """[Ci,KH,KW,Co] copy of a conv weight for the data gradient, shared by the two views' backward passes."""
            sh = T._shadow_perm.get(name)
            if sh is not None:                # made once per train_step for the whole trunk (ResNetTrunk.build_shadows)
                return sh
            key = (name, torch.cuda.current_stream().cuda_stream)     # (two-stream view overlap: one copy per stream)
            t = wt_cache.get(key)
            if t is None:
                t = wt_cache[key] = K.permute(p[name + ".weight"])
            return t

        # Weight gradients are off the critical chain (dgrad -> BN backward -> dgrad ...): they are issued on a side
        # stream so the MFMA-bound wgrad kernels overlap the HBM-bound BatchNorm-backward kernels of the main stream.
        main = torch.cuda.current_stream()
        side = T.wgrad_stream() if _WGRAD_SIDE_STREAM else None

        # Inside a train_step the passes that share this trunk's weights (the two views) sum their parameter gradients HERE, one
        # multi-tensor add per residual stage, instead of handing ~160 tensors per pass to the autograd engine (one add kernel
        # each: 319 launches per step) -- and a stage's gradients are complete, delivered to .grad and reported to the gradient
        # exchange (ResNetTrunk.grad_sink <- dist.GradSync) as soon as the LAST pass has finished that stage, so encoder buckets
        # are exchanged during backward instead of after the last trunk node returns.  The sums are the engine's: first + second.
        st = ctx.step if (ctx.step is not None and ctx.step is T._step) else None
        stash_mode = None
        if st is not None and _GRAD_STASH and st["nodes"] > 1 and cap is None:
            st["done"] += 1
            stash_mode = "last" if st["done"] >= st["nodes"] else "first"

        def flush(stage):
            """Stage `stage` ("layer4".."layer1", "stem") of this pass is done."""
            if stash_mode is None:
                return
            if side is not None:
                main.wait_stream(side)
            cur = torch.cuda.current_stream()
            names = [n for n in T.stage_params[stage] if grads.get(n) is not None]
            stash, ev = st["stash"], st["events"].pop(stage, None)
            if ev is not None and ev[0] != cur.cuda_stream:
                cur.wait_event(ev[1])               # the earlier pass ran this stage on another stream (view overlap)
            dst, src = [], []
            for n in names:
                g = grads.pop(n)                    # (popped: the node returns None for it, the engine sees no gradient)
                d = stash.get(n)
                if d is None:
                    pg = T.get(n).grad              # a gradient tensor that is already there: DP bucket view / accumulation
                    if pg is None:
                        stash[n] = g                # the first arrival's tensor becomes the accumulator
                        continue
                    d = stash[n] = pg
                dst.append(d); src.append(g)
            if dst:
                torch._foreach_add_(dst, src)
            if stash_mode != "last":
                st["events"][stage] = (cur.cuda_stream, cur.record_event())
                return
            ready = []
            for n in names:
                prm, d = T.get(n), stash.pop(n)
                if ev is not None and ev[0] != cur.cuda_stream:
                    d.record_stream(cur)
                if prm.grad is not d:
                    prm.grad = d
                ready.append(prm)
            if T.grad_sink is not None and ready:
                T.grad_sink(ready)

        def conv_bwd(name, dy, inp, stride, pad, need_dx=True, dx_out=None, accumulate=False):
            w = p[name + ".weight"]
            if side is None:
                grads[name + ".weight"] = K.conv_wgrad(dy, inp, tuple(w.shape), stride, pad)
                if cap is not None:
                    cap[name].update(d_raw=dy, dW=grads[name + ".weight"],
                                     dx_before=dx_out.clone() if (accumulate and dx_out is not None) else None)
                if not need_dx:
                    return None
                dx = K.conv_dgrad(dy, wt_of(name), tuple(inp.shape), stride, pad, out=dx_out, accumulate=accumulate)
                if cap is not None:
                    cap[name]["dx_after"] = dx.clone()
                return dx
            # Side-stream schedule: [join the previous wgrad] -> dgrad (alone on the GPU: its timing stays clean) ->
            # wgrad on the side stream, which then runs beside the NEXT layer's BatchNorm-backward kernels (MFMA-bound
            # beside HBM-bound) until the next dgrad joins it.
            main.wait_stream(side)
            dx = None
            if need_dx:
                dx = K.conv_dgrad(dy, wt_of(name), tuple(inp.shape), stride, pad, out=dx_out, accumulate=accumulate)
            side.wait_event(main.record_event())
            with torch.cuda.stream(side):
                dw = K.conv_wgrad(dy, inp, tuple(w.shape), stride, pad)
            dy.record_stream(side); inp.record_stream(side); dw.record_stream(main)
            grads[name + ".weight"] = dw
            return dx

        def bn_bwd(name, dy, mask, raw, st, want_dres=False):
            d_raw, dg, db, dres = K.bn_bwd(dy, mask, raw, st[0], st[1], p[name + ".weight"], want_dres)
            grads[name + ".weight"] = dg
            grads[name + ".bias"] = db
            if cap is not None:
                cap["bwd:" + name] = dict(dout=dy.clone(), dgamma=dg, dbeta=db, d_raw=d_raw,
                                          dres=dres.clone() if dres is not None else None)
            return d_raw, dres

        # ---- fused units
        def fin_bwd(bn_name, part, chunks, planes, raw, fc):
            C = raw.shape[-1]
            bc, dg, db = _bcoef_from_partials(part, chunks, planes, raw.numel() // C, p[bn_name + ".weight"], fc)
            grads[bn_name + ".weight"] = dg
            grads[bn_name + ".bias"] = db
            return bc

        def fwgrad(name, g, raw, bc, xin, x_fc, stride, pad, d=None):
            w = p[name + ".weight"]
            if d is not None:       # d = the unit's materialised d_raw (draw_sep)
                grads[name + ".weight"] = (K.conv_wgrad(d, xin, tuple(w.shape), stride, pad) if x_fc is None else
                                           K.wgrad_bn(d, None, None, xin, x_fc, tuple(w.shape), stride, pad))
                return
            grads[name + ".weight"] = K.wgrad_bn(g, raw, bc, xin, x_fc, tuple(w.shape), stride, pad)

        def draw_of(g, raw, bc):
            C_ = raw.shape[-1]
            d_ = torch.empty_like(raw)
            ops.call_timed_bytes("bn_draw", raw.numel() * 3 * K.elt, K.draw_name, P(g), P(raw), P(bc), P(d_), raw.numel() // C_, C_)
            return d_

        def fcap(conv_name, bn_name, g, raw, bc, dres=None):
            if cap is not None:
                dr = _dbg_draw(g, raw, bc)
                cap["bwd:" + bn_name] = dict(dout=g.clone(), dgamma=grads[bn_name + ".weight"], dbeta=grads[bn_name + ".bias"],
                                             d_raw=dr, dres=dres, masked=True)
                cap[conv_name].update(d_raw=dr, dW=grads[conv_name + ".weight"])

        def fdgrad(name, g, raw, bc, x_shape, stride, pad, out=None, accumulate=False, ep=None, ep_keep=None, d=None):
            before = out.clone() if (cap is not None and accumulate) else None
            if d is not None and ep is None:
                r = K.conv_dgrad(d, wt_of(name), tuple(x_shape), stride, pad, out=out, accumulate=accumulate)
            elif d is not None:
                r = K.dgrad_bn(d, None, None, wt_of(name), tuple(x_shape), stride, pad, out=out, accumulate=accumulate, ep=ep)
            else:
                r = K.dgrad_bn(g, raw, bc, wt_of(name), tuple(x_shape), stride, pad, out=out, accumulate=accumulate, ep=ep)
            if cap is not None:
                cap[name].update(dx_before=before, dx_after=(r if ep is None else r[0]).clone(),
                                 dx_keep=ep_keep() if (ep is not None and ep_keep is not None) else None)
            return r

        def lower_ep(bi):
            """Epilogue operands for the BatchNorm(+ReLU) that produced the input of block bi (the block below's last unit),
            when that block is fused: (raw, sign bytes, fcoef, relu)."""
            if bi == 0:
                return None, None
            lo = saved[T.blocks[bi - 1]["name"]]
            if not lo.get("fused"):
                return None, None
            raw_lo, fc_lo = (lo["c3"], lo["f3"]) if T.kind == "bottleneck" else (lo["c2"], lo["f2"])
            kl = lo["kl"]
            keep = lambda: _mask_to_bool(kl, raw_lo.shape)
            return (raw_lo, kl, fc_lo, True), keep

        def recompute_keep(raw, fc):     # (tests) the pre-activation whose sign the epilogue re-derives
            return lambda: _dbg_pre(raw, fc)

        # ---- recompute mode (T.recompute_out): block outputs are rebuilt from the stored raw tensors, one elementwise pass
        # each (the forward's own edrl_bn_apply_res_f32, bit-identical), a residual stage at a time; out_j is dropped as soon
        # as block j+1 is done.
        out_cache = {}

        def get_out(j):
            t = out_cache.get(j)
            if t is None:
                rj = saved[T.blocks[j]["name"]]
                cl, fl = (rj["c3"], rj["f3"]) if T.kind == "bottleneck" else (rj["c2"], rj["f2"])
                res, rfc = (rj["cd"], rj["fd"]) if T.blocks[j]["downsample"] else (get_x(j), None)
                t = torch.empty_like(cl)
                Cl = cl.shape[-1]
                kl = None
                if rj.get("kl") is None:          # the forward pass did not keep the sign bytes (recompute mode): rebuilt here
                    kl = rj["kl"] = torch.empty((cl.numel() // Cl, Cl // 4), device=cl.device, dtype=torch.uint8)
                ops.call_timed_bytes("bn_apply_res", cl.numel() * (3 * K.elt + (0.25 if kl is not None else 0.0)), K.apply_res_name,
                                     P(cl), P(fl), P(res), P(rfc), P(t), P(kl), cl.numel() // Cl, Cl, 1)
                out_cache[j] = t
            return t

        def get_x(j):
            xj = saved[T.blocks[j]["name"]]["x"]
            return xj if xj is not None else get_out(j - 1)

        grad_in = ("plain", K.grad_in(dout))
        for bi in range(len(T.blocks) - 1, -1, -1):
            blk = T.blocks[bi]
            pre, s = blk["name"], blk["stride"]
            rec = saved[pre]
            xin = get_x(bi)                      # (recompute mode: rebuilds the block below's output AND its sign bytes)
            ep_lo, keep_lo = lower_ep(bi)
            out_cache.pop(bi, None)
            if rec.get("fused"):
                bott = T.kind == "bottleneck"
                dsep = bott and not rec.get("wide") and rec["c1"].shape[-1] >= K.draw_sep_min_planes
                last, last_bn = (pre + ".conv3", pre + ".bn3") if bott else (pre + ".conv2", pre + ".bn2")
                cl, fl = (rec["c3"], rec["f3"]) if bott else (rec["c2"], rec["f2"])
                if grad_in[0] == "plain":
                    gl, part, chunks, planes = _bn_bwd_reduce(K, grad_in[1], rec["kl"], cl, fl, want_g=True)
                else:
                    _, gl, part, chunks, planes = grad_in
                bl = fin_bwd(last_bn, part, chunks, planes, cl, fl)
                c1, f1 = rec["c1"], rec["f1"]
                if rec.get("wide"):
                    # materialised d_raw per unit (one pass), plain weight / data gradients on the LDS-DMA cores, the BatchNorm
                    # backward statistics from the data gradients' epilogues (sign bytes of the BatchNorm below)
                    def draw(g_, raw_, bc_):
                        Cc = raw_.shape[-1]
                        d_ = torch.empty_like(raw_)
                        ops.call_timed_bytes("bn_draw", raw_.numel() * 3 * K.elt, K.draw_name, P(g_), P(raw_), P(bc_), P(d_),
                                             raw_.numel() // Cc, Cc)
                        return d_

                    def wg(name, d_, inp, st_, pd_):
                        grads[name + ".weight"] = K.conv_wgrad(d_, inp, tuple(p[name + ".weight"].shape), st_, pd_)

                    c2, f2 = rec["c2"], rec["f2"]
                    d3 = draw(gl, cl, bl)
                    wg(last, d3, rec["a2"], 1, 0)
                    g2, part, chunks = K.dgrad_bn(d3, None, None, wt_of(last), tuple(c2.shape), 1, 0, ep=(c2, rec["k2"], f2, True))
                    del d3
                    b2 = fin_bwd(pre + ".bn2", part, chunks, 2, c2, f2)
                    d2 = draw(g2, c2, b2)
                    del g2
                    wg(pre + ".conv2", d2, rec["a1"], s, 1)
                    g1, part, chunks = K.dgrad_bn(d2, None, None, wt_of(pre + ".conv2"), tuple(c1.shape), s, 1,
                                                  ep=(c1, rec["k1"], f1, True))
                    del d2
                    b1 = fin_bwd(pre + ".bn1", part, chunks, 2, c1, f1)
                    d1 = draw(g1, c1, b1)
                    del g1
                    wg(pre + ".conv1", d1, xin, 1, 0)
                    if blk["downsample"]:
                        cd, fd = rec["cd"], rec["fd"]
                        _, partd, chunksd, planesd = _bn_bwd_reduce(K, gl, None, cd, fd, want_g=False)
                        bd = fin_bwd(pre + ".downsample.1", partd, chunksd, planesd, cd, fd)
                        dd = draw(gl, cd, bd)
                        wg(pre + ".downsample.0", dd, xin, s, 0)
                        dx = K.conv_dgrad(dd, wt_of(pre + ".downsample.0"), tuple(xin.shape), s, 0)     # full cover (zero fill off-lattice)
                        del dd
                    else:
                        dx = gl
                    if ep_lo is not None:
                        r = K.dgrad_bn(d1, None, None, wt_of(pre + ".conv1"), tuple(xin.shape), 1, 0, out=dx, accumulate=True, ep=ep_lo)
                        grad_in = ("masked", r[0], r[1], r[2], 2)
                    else:
                        grad_in = ("plain", K.conv_dgrad(d1, wt_of(pre + ".conv1"), tuple(xin.shape), 1, 0, out=dx, accumulate=True))
                    del rec, saved[pre]
                    if bi == 0 or T.blocks[bi - 1]["name"].split(".")[0] != pre.split(".")[0]:
                        flush(pre.split(".")[0])
                    continue
                if bott:
                    c2, f2 = rec["c2"], rec["f2"]
                    if cap is None and K.conv3_bwd_ok(c2.shape[0], c2.shape[1], c2.shape[2], c2.shape[3], cl.shape[3]):
                        # expanding 1x1 layer of the first stage: both gradients from one pass over (gl, cl)
                        grads[last + ".weight"], g2, part, chunks = K.conv3_bwd(gl, cl, bl, c2, f2, wt_of(last))
                    else:
                        d3 = draw_of(gl, cl, bl) if dsep else None
                        fwgrad(last, gl, cl, bl, c2, f2, 1, 0, d=d3)
                        fcap(last, last_bn, gl, cl, bl, dres=gl.clone() if cap is not None else None)
                        g2, part, chunks = fdgrad(last, gl, cl, bl, c2.shape, 1, 0, ep=(c2, None, f2, True),
                                                  ep_keep=recompute_keep(c2, f2), d=d3)
                        del d3
                    b2 = fin_bwd(pre + ".bn2", part, chunks, 2, c2, f2)
                    planes1 = 2
                    if rec.get("mid_sep"):
                        if rec["a1"] is None:      # recompute mode: bit-identical to the forward's pass
                            C1 = c1.shape[-1]
                            M1 = c1.numel() // C1
                            rec["a1"] = torch.empty_like(c1)
                            rec["k1"] = torch.empty((M1, C1 // 4), device=c1.device, dtype=torch.uint8)
                            mu_, sh_ = _act_coef(K, f1)
                            ops.call_timed_bytes("bn_apply", M1 * C1 * (2 * K.elt + 0.25), "edrl_bn_apply_mx", P(c1), K.mx, P(mu_), P(f1[2]),
                                                 P(sh_), None, P(rec["a1"]), K.mx, P(rec["k1"]), M1, C1, 1)
                        C2c = c2.shape[-1]
                        d2 = torch.empty_like(c2)
                        ops.call_timed_bytes("bn_draw", c2.numel() * 3 * K.elt, K.draw_name, P(g2), P(c2), P(b2), P(d2), c2.numel() // C2c, C2c)
                        w2 = p[pre + ".conv2.weight"]
                        grads[pre + ".conv2.weight"] = K.conv_wgrad(d2, rec["a1"], tuple(w2.shape), s, 1)
                        # plain operand (d2 is materialised), epilogue as in the fused chain: masked with bn1's sign bytes,
                        # (sum g, sum g*x) partials out -- no standalone reduce pass over da1
                        g1, part, chunks = K.dgrad_bn(d2, None, None, wt_of(pre + ".conv2"), tuple(c1.shape), s, 1,
                                                      ep=(c1, rec["k1"], f1, True))
                        if cap is not None:
                            cap["bwd:" + pre + ".bn2"] = dict(dout=g2.clone(), dgamma=grads[pre + ".bn2.weight"],
                                                            dbeta=grads[pre + ".bn2.bias"], d_raw=d2, dres=None, masked=True)
                            cap[pre + ".conv2"].update(d_raw=d2, dW=grads[pre + ".conv2.weight"], dx_before=None, dx_after=g1.clone(),
                                                       dx_keep=_mask_to_bool(rec["k1"], c1.shape))
                        del d2
                    else:
                        fwgrad(pre + ".conv2", g2, c2, b2, c1, f1, s, 1)
                        fcap(pre + ".conv2", pre + ".bn2", g2, c2, b2)
                        g1, part, chunks = fdgrad(pre + ".conv2", g2, c2, b2, c1.shape, s, 1, ep=(c1, None, f1, True),
                                                  ep_keep=recompute_keep(c1, f1))
                    c1_stride, c1_pad = 1, 0
                else:
                    fwgrad(last, gl, cl, bl, c1, f1, 1, 1)
                    fcap(last, last_bn, gl, cl, bl, dres=gl.clone() if cap is not None else None)
                    g1, part, chunks = fdgrad(last, gl, cl, bl, c1.shape, 1, 1, ep=(c1, None, f1, True),
                                              ep_keep=recompute_keep(c1, f1))
                    c1_stride, c1_pad = s, 1
                b1 = fin_bwd(pre + ".bn1", part, chunks, planes1 if bott else 2, c1, f1)
                d1 = draw_of(g1, c1, b1) if dsep else None
                fwgrad(pre + ".conv1", g1, c1, b1, xin, None, c1_stride, c1_pad, d=d1)
                fcap(pre + ".conv1", pre + ".bn1", g1, c1, b1)
                # block-input gradient: (downsample branch | identity) first, conv1's data gradient accumulated last so that
                # its epilogue sees the complete gradient of the block below's output
                if blk["downsample"]:
                    cd, fd = rec["cd"], rec["fd"]
                    _, partd, chunksd, planesd = _bn_bwd_reduce(K, gl, None, cd, fd, want_g=False)
                    bd = fin_bwd(pre + ".downsample.1", partd, chunksd, planesd, cd, fd)
                    dd = draw_of(gl, cd, bd) if dsep else None
                    fwgrad(pre + ".downsample.0", gl, cd, bd, xin, None, s, 0, d=dd)
                    fcap(pre + ".downsample.0", pre + ".downsample.1", gl, cd, bd)
                    dx = fdgrad(pre + ".downsample.0", gl, cd, bd, xin.shape, s, 0, d=dd)      # full cover (zero fill off-lattice)
                    del dd
                else:
                    dx = gl
                r = fdgrad(pre + ".conv1", g1, c1, b1, xin.shape, c1_stride, c1_pad, out=dx, accumulate=True, ep=ep_lo,
                           ep_keep=keep_lo, d=d1)
                del d1
                grad_in = ("plain", r) if ep_lo is None else ("masked", r[0], r[1], r[2], 2)
                del rec, saved[pre]
                if bi == 0 or T.blocks[bi - 1]["name"].split(".")[0] != pre.split(".")[0]:
                    flush(pre.split(".")[0])
                continue
            # ---- separate-pass block
            if grad_in[0] == "masked":      # (not produced: a fused block masks only for a fused block below)
                raise RuntimeError("internal: masked gradient handed to an unfused block")
            dcur = grad_in[1]
            if T.kind == "bottleneck":
                d3, g = bn_bwd(pre + ".bn3", dcur, rec["kl"], rec["c3"], rec["sl"], want_dres=True)
            else:
                d3, g = bn_bwd(pre + ".bn2", dcur, rec["kl"], rec["c2"], rec["sl"], want_dres=True)
            # Block-input gradient dx = (main path: conv1's dgrad, written first, covers every pixel)
            #                         + (identity: g | downsample: its strided dgrad, accumulated afterwards so that
            #                            only the parity class it reaches is touched — no zero fill, no re-read).
            if T.kind == "bottleneck":
                da2 = conv_bwd(pre + ".conv3", d3, rec["a2"], 1, 0)
                d2, _ = bn_bwd(pre + ".bn2", da2, rec["k2"], rec["c2"], rec["s2"])
                da1 = conv_bwd(pre + ".conv2", d2, rec["a1"], s, 1)
                d1, _ = bn_bwd(pre + ".bn1", da1, rec["k1"], rec["c1"], rec["s1"])
                c1_stride, c1_pad = 1, 0
            else:
                da1 = conv_bwd(pre + ".conv2", d3, rec["a1"], 1, 1)
                d1, _ = bn_bwd(pre + ".bn1", da1, rec["k1"], rec["c1"], rec["s1"])
                c1_stride, c1_pad = s, 1
            if blk["downsample"]:
                dx = conv_bwd(pre + ".conv1", d1, xin, c1_stride, c1_pad)
                dd, _ = bn_bwd(pre + ".downsample.1", g, None, rec["cd"], rec["sd"])
                conv_bwd(pre + ".downsample.0", dd, xin, s, 0, dx_out=dx, accumulate=True)
                del dd
            else:
                dx = g
                conv_bwd(pre + ".conv1", d1, xin, c1_stride, c1_pad, dx_out=dx, accumulate=True)
            del g
            if ep_lo is not None:            # the block below is fused: hand it the masked gradient + partial sums
                lo_raw, lo_mask, lo_fc, _ = ep_lo
                gl, part, chunks, planes = _bn_bwd_reduce(K, dx, lo_mask, lo_raw, lo_fc, want_g=True)
                grad_in = ("masked", gl, part, chunks, planes)
            else:
                grad_in = ("plain", dx)
            del rec, saved[pre]
            if bi == 0 or T.blocks[bi - 1]["name"].split(".")[0] != pre.split(".")[0]:
                flush(pre.split(".")[0])
        dx = K.stem_bwd(T, saved.pop("stem"), p, grad_in[1], grads, cap, bn_bwd, conv_bwd, ctx.needs_x)
        flush("stem")
        if side is not None:
            main.wait_stream(side)
        return (None, dx) + tuple(grads.get(n) for n in T.param_names)


def _mask_to_bool(mask, shape):
    """(tests only) ReLU sign bytes [M, C/4] -> bool tensor of `shape` [..., C]"""
    C = shape[-1]
    m = mask.view(-1, C // 4, 1).to(torch.int32)
    sh = torch.arange(4, dtype=torch.int32, device=mask.device).view(1, 1, 4)
    return ((m >> sh) & 1).bool().view(shape)


# ------------------------------------------------------------------------------------------------ bf16 trunk (C2/C4)
def _conv_bn_fwd_bf16(inp, w, bn, stride, pad, relu, residual=None):
    """bf16 conv (fp32 accumulate, BN statistics from the accumulators) -> BN -> (+residual) -> (ReLU), all tensors bf16.
    -> (raw bf16, out bf16, mean, rstd, mask)."""
    raw, part, chunks = ops.conv2d_fwd_bf16(inp, _wbf16(w), stride, pad, stats=True)
    C = raw.shape[-1]
    M = raw.numel() // C
    dev = raw.device
    mean = torch.empty(C, device=dev, dtype=torch.float32)
    rstd = torch.empty_like(mean); scale = torch.empty_like(mean); shift = torch.empty_like(mean)
    gbytes = L.query("edrl_bn_finalize_group_ws_bytes", chunks, C)
    gws = torch.empty(max(gbytes // 8, 1), device=dev, dtype=torch.float64)
    L.call("edrl_bn_finalize_partials_f32", P(part), chunks, 128, M, C, P(bn["weight"]), P(bn["bias"]),
           P(bn["running_mean"]), P(bn["running_var"]), float(bn["momentum"]), float(bn["eps"]), P(mean), P(rstd),
           P(scale), P(shift), P(gws), gbytes)
    out = torch.empty_like(raw)
    mask = torch.empty((M, C // 4), device=dev, dtype=torch.uint8) if relu else None
    L.call("edrl_bn_apply_mx", P(raw), 1, P(mean), P(scale), P(shift), P(residual), P(out), 1, P(mask), M, C,
           1 if relu else 0)
    return raw, out, mean, rstd, mask


def _bn_bwd_mx(dout, mask, raw, mean, rstd, gamma, want_dres):
    """BN(+ReLU) backward with bf16 gradients; d_raw takes the storage type of `raw` (bf16, or fp32 for the stem)."""
    C = raw.shape[-1]
    M = raw.numel() // C
    dev = raw.device
    d_raw = torch.empty_like(raw)
    dgamma = torch.empty(C, device=dev, dtype=torch.float32)
    dbeta = torch.empty_like(dgamma)
    dres = torch.empty(raw.shape, device=dev, dtype=torch.bfloat16) if want_dres else None
    ws, nbytes = _bn_ws(M, C, dev, extra=2 * C * 4)
    L.call("edrl_bn_bwd_mx", P(dout), 1, P(mask), P(raw), 1 if raw.dtype == torch.bfloat16 else 0, P(mean), P(rstd),
           P(gamma), P(dgamma), P(dbeta), P(d_raw), P(dres), M, C, P(ws), nbytes)
    return d_raw, dgamma, dbeta, dres


class ResNetTrunk(nn.Module):
    """ResNet-18/34/50 trunk (no fc), NHWC fp32, HIP kernels only."""

    def __init__(self, depth=50, in_ch=3, dtype="fp32"):
        super().__init__()
        assert dtype in ("fp32", "bf16")
        self.compute_dtype = dtype
        kind, layers, expansion = _CFG[depth]
        self.kind, self.depth, self.in_ch = kind, depth, in_ch
        self.in_ch_padded = in_ch if in_ch == 1 else (in_ch + 3) // 4 * 4
        self.blocks = []
        self._bn_names = []
        self._add_conv("conv1", 64, 7, self.in_ch_padded, real_ci=in_ch)
        self._add_bn("bn1", 64)
        inpl = 64
        for li, (planes, nblk) in enumerate(zip([64, 128, 256, 512], layers)):
            for bi in range(nblk):
                stride = 2 if (bi == 0 and li > 0) else 1
                pre = f"layer{li + 1}.{bi}"
                outp = planes * expansion
                ds = (stride != 1) or (inpl != outp)
                if kind == "bottleneck":
                    self._add_conv(pre + ".conv1", planes, 1, inpl); self._add_bn(pre + ".bn1", planes)
                    self._add_conv(pre + ".conv2", planes, 3, planes); self._add_bn(pre + ".bn2", planes)
                    self._add_conv(pre + ".conv3", outp, 1, planes); self._add_bn(pre + ".bn3", outp, zero=False)
                else:
                    self._add_conv(pre + ".conv1", planes, 3, inpl); self._add_bn(pre + ".bn1", planes)
                    self._add_conv(pre + ".conv2", planes, 3, planes); self._add_bn(pre + ".bn2", planes)
                if ds:
                    self._add_conv(pre + ".downsample.0", outp, 1, inpl); self._add_bn(pre + ".downsample.1", outp)
                self.blocks.append({"name": pre, "stride": stride, "downsample": ds})
                inpl = outp
        self.out_channels = inpl
        self.param_names = [n for n, _ in self.named_parameters()]
        self._capture = None
        self._capture_seq = None
        self._wt_cache = {}
        self._step = None            # per-train_step state (step_begin / step_end)
        self._shadow_perm = {}       # conv name -> permuted data-gradient operand, valid inside a train_step (build_shadows)
        self._shadow_state = None
        self.grad_sink = None        # callable(list of parameters whose .grad is complete): dist.GradSync.params_ready
        self.stage_params = {}
        for n in self.param_names:
            st = n.split(".")[0]
            self.stage_params.setdefault(st if st.startswith("layer") else "stem", []).append(n)
        self.kernels = _KBF16 if dtype == "bf16" else _K32
        self._register_state_dict_hook(self._save_hook)
        self._register_load_state_dict_pre_hook(self._load_pre_hook)
        # True: do not keep the residual blocks' outputs for backward (rebuilt there from the raw conv outputs: one extra
        # elementwise pass per block, -30 % activation memory) -- what lets BASELINE.json's B=64/GPU fp32 shapes fit 288 GB
        self.recompute_out = os.environ.get("EDRL_RECOMPUTE_OUT", "0") == "1"

    # parameters are registered under dotted torchvision-style names with '.' -> '__' for attribute safety
    def _reg(self, name, tensor, buffer=False):
        key = name.replace(".", "__")
        if buffer:
            self.register_buffer(key, tensor)
        else:
            self.register_parameter(key, nn.Parameter(tensor))

    def _add_conv(self, name, co, k, ci, real_ci=None):
        w = torch.empty(co, k, k, ci)
        fan_out = co * k * k  # kaiming_normal_(mode='fan_out', nonlinearity='relu')
        w.normal_(0.0, math.sqrt(2.0 / fan_out))
        if real_ci is not None and real_ci < ci:
            w[..., real_ci:] = 0.0
        self._reg(name + ".weight", w)

    def _add_bn(self, name, c, zero=False):
        self._reg(name + ".weight", torch.zeros(c) if zero else torch.ones(c))
        self._reg(name + ".bias", torch.zeros(c))
        self._reg(name + ".running_mean", torch.zeros(c), buffer=True)
        self._reg(name + ".running_var", torch.ones(c), buffer=True)
        self._reg(name + ".num_batches_tracked", torch.zeros((), dtype=torch.long), buffer=True)
        self._bn_names.append(name)

    def named_parameters(self, *a, **k):
        for n, v in super().named_parameters(*a, **k):
            yield n.replace("__", "."), v

    def get(self, name):
        return getattr(self, name.replace(".", "__"))

    def bn_dict(self, name, p):
        sc = getattr(self, "_scratch_running", None)
        if sc is not None:      # a pass running concurrently with another one: see begin_scratch_running()
            rm, rv = sc[name]
        else:
            rm, rv = self.get(name + ".running_mean"), self.get(name + ".running_var")
        return {"weight": p[name + ".weight"], "bias": p[name + ".bias"], "running_mean": rm, "running_var": rv,
                "momentum": 0.1, "eps": 1e-5}

    # Two training passes that run CONCURRENTLY on two streams (train.train_step, EDRL_VIEW_STREAM=1) must still update
    # the running statistics as if they had run one after the other (fusion_train.py:189-194: low view, then high
    # view).  The update is linear, r <- (1-m) r + m s: the second pass accumulates m*s into zeroed scratch buffers and
    # merge_scratch_running() folds them in after both passes have been joined: r <- (1-m) r_after_pass1 + m s2.
    def begin_scratch_running(self):
        bufs = getattr(self, "_scratch_bufs", None)
        if bufs is None:
            bufs = self._scratch_bufs = {n: (torch.zeros_like(self.get(n + ".running_mean")),
                                             torch.zeros_like(self.get(n + ".running_var"))) for n in self._bn_names}
        else:
            torch._foreach_zero_([t for pair in bufs.values() for t in pair])
        self._scratch_running = bufs

    def end_scratch_running(self):
        self._scratch_running = None

    def merge_scratch_running(self):
        bufs = self._scratch_bufs
        run = [self.get(n + s) for n in self._bn_names for s in (".running_mean", ".running_var")]
        torch._foreach_mul_(run, 0.9)
        torch._foreach_add_(run, [t for n in self._bn_names for t in bufs[n]])
        # the concurrent pass did not touch num_batches_tracked (bump_batches_tracked): its increment is applied here, in
        # order on the joining stream, so the counter advances by exactly 2 per step whatever the streams' interleaving
        torch._foreach_add_([self.get(n + ".num_batches_tracked") for n in self._bn_names], 1)

    # ---- one optimisation step (train.train_step brackets it): weight shadows + the gradient stash of _TrunkFn.backward
    def step_begin(self):
        self._step = {"nodes": 0, "done": 0, "stash": {}, "events": {}}
        if _WEIGHT_SHADOWS:
            self.build_shadows()

    def stash_active(self):
        """True while the passes of the current train_step sum this trunk's parameter gradients themselves (_TrunkFn.backward)."""
        st = self._step
        return st is not None and _GRAD_STASH and st["nodes"] > 1 and self._capture is None and self._capture_seq is None

    def step_end(self):
        st, self._step = self._step, None
        self.drop_shadows()
        if st is not None and st["stash"]:
            raise RuntimeError(f"ResNetTrunk: {len(st['stash'])} parameter gradients of an unfinished backward were left in the "
                               "step's stash (a forward pass of this step never ran its backward)")

    def build_shadows(self):
        """bf16 forward operands + permuted data-gradient operands of every conv weight but the stem's, one launch
        (edrl_weight_shadows_multi).  Buffers and the device table are allocated once and reused while the parameters stay put."""
        import struct
        names = [n[:-7] for n in self.param_names if n.endswith(".weight") and n != "conv1.weight" and self.get(n).dim() == 4]
        ws = [self.get(n + ".weight") for n in names]
        if not ws or not ws[0].is_cuda:
            return
        key = tuple(w.data_ptr() for w in ws)
        sh = self._shadow_state
        if sh is None or sh["key"] != key:
            bf16 = self.compute_dtype == "bf16"
            dev = ws[0].device
            ce = L.lib().fn["edrl_adam_chunk_elems"]()
            recs, chunks, cast, perm = [], [], {}, {}
            for ti, (n, w) in enumerate(zip(names, ws)):
                Co, KH, KW, Ci = w.shape
                c = torch.empty(w.shape, device=dev, dtype=torch.bfloat16) if bf16 else None
                pm = torch.empty((Ci, KH, KW, Co), device=dev, dtype=torch.bfloat16 if bf16 else torch.float32)
                recs.append(struct.pack("QQQiiiiq", w.data_ptr(), c.data_ptr() if c is not None else 0, pm.data_ptr(), Co, KH * KW,
                                        Ci, 1 if bf16 else 0, w.numel()))
                chunks.extend((ti, k) for k in range((w.numel() + ce - 1) // ce))
                if c is not None:
                    cast[w.data_ptr()] = c
                perm[n] = pm
            sh = self._shadow_state = {
                "key": key, "cast": cast, "perm": perm, "n": len(names), "n_chunks": len(chunks),
                "table": torch.frombuffer(bytearray(b"".join(recs)), dtype=torch.uint8).to(dev),
                "chunks": torch.tensor(chunks, dtype=torch.int32).reshape(-1, 2).to(dev)}
        L.call("edrl_weight_shadows_multi", sh["table"].data_ptr(), sh["n"], sh["chunks"].data_ptr(), sh["n_chunks"])
        _SHADOW_CAST.update(sh["cast"])
        self._shadow_perm = sh["perm"]

    def drop_shadows(self):
        sh = self._shadow_state
        if sh is not None:
            for k in sh["cast"]:
                _SHADOW_CAST.pop(k, None)
        self._shadow_perm = {}

    def wgrad_stream(self):
        s = getattr(self, "_wgrad_stream", None)
        if s is None:
            s = self._wgrad_stream = torch.cuda.Stream()
        return s

    def bump_batches_tracked(self):
        if getattr(self, "_scratch_running", None) is not None:
            return          # a pass running beside another one on a second stream: counted in merge_scratch_running()
        torch._foreach_add_([self.get(n + ".num_batches_tracked") for n in self._bn_names], 1)

    # ---- checkpoint key / layout compatibility: state_dict() exposes torchvision-style names ("layer1.0.conv1.weight",
    # "bn1.running_mean") and conv weights in torch's [Co,Ci,KH,KW] layout without the stem's zero padding channel, so that
    # NCHW ResNet checkpoints load and saved ones are readable elsewhere; internally parameters stay [Co,KH,KW,Ci] under
    # attribute-safe names ('.' -> '__').  Checkpoints written with the internal names / layout still load.
    def _save_hook(self, module, sd, prefix, local_meta):
        for key in [k for k in sd if k.startswith(prefix) and "__" in k[len(prefix):]]:
            name = key[len(prefix):].replace("__", ".")
            t = sd.pop(key)
            if t.dim() == 4:
                t = t.permute(0, 3, 1, 2)
                if name == "conv1.weight" and self.in_ch_padded != self.in_ch:
                    t = t[:, :self.in_ch]
                t = t.contiguous()
            sd[prefix + name] = t
        return sd

    def _load_pre_hook(self, sd, prefix, local_meta, strict, missing, unexpected, errors):
        own = {n.replace("__", "."): n for n in list(self._parameters) + list(self._buffers)}
        for name, internal in own.items():
            key = prefix + name
            if key not in sd or name == internal:
                continue
            t = sd.pop(key)
            if t.dim() == 4:                                  # [Co,Ci,KH,KW] -> [Co,KH,KW,Ci] (+ the stem's zero channel)
                t = t.permute(0, 2, 3, 1)
                if name == "conv1.weight" and t.shape[-1] == self.in_ch and self.in_ch_padded != self.in_ch:
                    t = torch.cat([t, t.new_zeros(*t.shape[:-1], self.in_ch_padded - self.in_ch)], dim=-1)
                t = t.contiguous()
            sd[prefix + internal] = t

    def forward(self, x_nhwc):
        if not self.training:
            return self.forward_eval(x_nhwc)
        params = [self.get(n) for n in self.param_names]
        return _TrunkFn.apply(self, x_nhwc, *params)

    @torch.no_grad()
    def forward_eval(self, x):
        """Inference forward (running-stat BatchNorm, nothing saved): same HIP conv / BN-apply / pooling kernels."""
        x = ops._chk(x, "encoder input")

        def bn(name, raw, relu, residual=None):
            return ops.batchnorm_eval(raw, self.get(name + ".running_mean"), self.get(name + ".running_var"),
                                      self.get(name + ".weight"), self.get(name + ".bias"), 1e-5, relu, residual)

        def conv(name, inp, stride, pad):
            return ops.conv2d_fwd(inp, self.get(name + ".weight"), stride=stride, pad=pad)

        a0 = bn("bn1", conv("conv1", x, 2, 3), True)
        N, H, W, C = a0.shape
        Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
        cur = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.float32)
        idx = torch.empty((N, Ho, Wo, C), device=x.device, dtype=torch.uint8)
        L.call("edrl_maxpool3x3s2_fwd_f32", P(a0), P(cur), P(idx), N, H, W, C)
        for blk in self.blocks:
            pre, s = blk["name"], blk["stride"]
            if self.kind == "bottleneck":
                o = bn(pre + ".bn1", conv(pre + ".conv1", cur, 1, 0), True)
                o = bn(pre + ".bn2", conv(pre + ".conv2", o, s, 1), True)
                last, last_bn = conv(pre + ".conv3", o, 1, 0), pre + ".bn3"
            else:
                o = bn(pre + ".bn1", conv(pre + ".conv1", cur, s, 1), True)
                last, last_bn = conv(pre + ".conv2", o, 1, 1), pre + ".bn2"
            idn = bn(pre + ".downsample.1", conv(pre + ".downsample.0", cur, s, 0), False) if blk["downsample"] else cur
            cur = bn(last_bn, last, True, residual=idn)
        return cur


class FundusEncoder(nn.Module):
    """2D fundus encoder slot: [B,3,H,W] (NCHW, as the loader emits, data_harvard.py:830-841)
    -> (tokens [B, (H/32)*(W/32), token_dim], pooled [B, token_dim])."""

    def __init__(self, depth=50, token_dim=1024, dtype="fp32"):
        super().__init__()
        self.trunk = ResNetTrunk(depth, in_ch=3, dtype=dtype)
        c = self.trunk.out_channels
        self.token_proj = nn.Linear(c, token_dim)

    def forward(self, x):
        ops._chk(x, "fundus")
        B, C, H, W = x.shape
        cp = self.trunk.in_ch_padded
        xh = torch.empty((B, H, W, cp), device=x.device, dtype=torch.float32)
        L.call("edrl_nchw_to_nhwc_f32", P(x), P(xh), B, C, H, W, cp)
        f = self.trunk(xh)                                  # [B,h,w,C]
        tok = f.view(B, f.shape[1] * f.shape[2], f.shape[3])
        tokens = ops.linear(tok, self.token_proj.weight, self.token_proj.bias)
        return tokens, ops.mean_axis1(tokens)


class OCTSliceEncoder(nn.Module):
    """OCT slice-stack encoder slot: [B,1,S,H,W] -> (tokens [B, S, token_dim], pooled [B, token_dim]).
    The S slices run through the 2D trunk as a batch of B*S single-channel images."""

    def __init__(self, depth=50, token_dim=768, dtype="fp32"):
        super().__init__()
        self.trunk = ResNetTrunk(depth, in_ch=1, dtype=dtype)
        c = self.trunk.out_channels
        self.token_proj = nn.Linear(c, token_dim)

    def forward(self, x):
        ops._chk(x, "oct")
        B, C, S, H, W = x.shape
        assert C == 1
        xh = x.view(B * S, H, W, 1)                         # single channel: NCHW == NHWC
        f = self.trunk(xh)                                  # [B*S,h,w,C]
        pooled = ops.mean_axis1(f.view(B * S, f.shape[1] * f.shape[2], f.shape[3]))   # global avg pool
        tokens = ops.linear(pooled.view(B, S, -1), self.token_proj.weight, self.token_proj.bias)
        return tokens, ops.mean_axis1(tokens)
