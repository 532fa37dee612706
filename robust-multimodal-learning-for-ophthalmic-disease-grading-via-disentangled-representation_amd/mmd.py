"""MK_MMD(source, target, kernel_mul=2.0, kernel_num=5) — drop-in for code/MMD.py:46-74.

Gram matrix on the fp32 MFMA GEMM, fused RBF/bandwidth/quadrant reduction and the full backward
(through the data-dependent bandwidth) in mmd.hip.  Gradients flow to both arguments, as
fusion_train.py:198 requires.
"""
from . import ops


def MK_MMD(source, target, kernel_mul=2.0, kernel_num=5):
    return ops.mk_mmd(source, target, kernel_mul, kernel_num)
