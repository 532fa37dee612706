"""MI355X-native (gfx950) implementation of the EDRL training hot path.

Drop-in surface:
    MedFusion(classes, modalties, classifiers_dims, args).forward(X, y, epoch) -> (pred, loss, combine_features)
    MK_MMD(source, target, kernel_mul=2.0, kernel_num=5) -> 0-d tensor
    train_step / train  (the loop body of fusion_train.py:176-225)
Compute lives in libedrl_hip.so (hand-written HIP, C-ABI in include/edrl_hip.h); importing this
package never builds or falls back: a missing library raises on first use.
"""
import os as _os

# Hardware queues per process (ROCclr default 4).  A data-parallel step runs on the compute stream, the second view's stream
# (train.train_step), GradSync's communication stream and RCCL's internal one; with 4 queues the two compute streams share a queue
# as soon as a process group exists and the two-view overlap is lost (C1, 1-rank RCCL group: 74.2 -> 77.0 images/s with 8).  The
# runtime reads the variable when HIP starts: effective if this import precedes the process's first GPU call (bench.py and the
# launch line of INTEGRATION.md set it in the environment as well); dist.init_process_group warns when it came too late.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib, ops
from . import custom_ops          # registers torch.ops.edrl.* (torch.library schemas over the same C-ABI launchers)
from .mmd import MK_MMD, compute_js_divergence, compute_kl_divergence
from .medfusion import MedFusion, EPRL, PoE, DILR, AttentionModel, off_diagonal
from .encoders import ResNetTrunk, FundusEncoder, OCTSliceEncoder
from .encoders3d import ResNet3DTrunk, OCTVolumeEncoder
from .train import train_step, train, val, synthetic_batch, device_twin_views, set_view_overlap, view_overlap, DevicePrefetcher
from . import dist
from .dist import GradSync, broadcast_parameters
from .optim import FusedAdam

__all__ = ["MedFusion", "EPRL", "PoE", "DILR", "AttentionModel", "off_diagonal", "MK_MMD", "compute_js_divergence", "compute_kl_divergence", "ResNetTrunk",
           "FundusEncoder", "OCTSliceEncoder", "ResNet3DTrunk", "OCTVolumeEncoder", "train_step", "train", "val", "synthetic_batch", "device_twin_views", "set_view_overlap", "view_overlap", "DevicePrefetcher", "ops", "GradSync",
           "broadcast_parameters", "FusedAdam"]
