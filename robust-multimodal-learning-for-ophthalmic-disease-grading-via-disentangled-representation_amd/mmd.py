"""MK_MMD(source, target, kernel_mul=2.0, kernel_num=5) — drop-in for code/MMD.py:46-74.

Gram matrix on the fp32 MFMA GEMM, fused RBF/bandwidth/quadrant reduction and the full backward
(through the data-dependent bandwidth) in mmd.hip.  Gradients flow to both arguments, as
fusion_train.py:198 requires.
"""
from . import ops


def MK_MMD(source, target, kernel_mul=2.0, kernel_num=5):
    return ops.mk_mmd(source, target, kernel_mul, kernel_num)


def compute_kl_divergence(p, m):
    """code/MMD.py:92-95."""
    return ops.KlRowsFn.apply(p, m)


def compute_js_divergence(p, q):
    """code/MMD.py:76-90: 0.5 * (KL(p||m) + KL(q||m)), m = 0.5 * (p + q) (its call site, fusion_train.py:207, is commented)."""
    m = ops.AxpbyFn.apply(p, q, 0.5, 0.5)
    return ops.scalar_mix([0.5, 0.5], [compute_kl_divergence(p, m), compute_kl_divergence(q, m)])
