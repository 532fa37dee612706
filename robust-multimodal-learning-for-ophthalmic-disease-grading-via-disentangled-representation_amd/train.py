"""The optimisation step of the reference driver, `fusion_train.train()` (fusion_train.py:166-265).

Call order reproduced exactly (fusion_train.py:189-224):
    optimizer.zero_grad() -> model(low view) -> model(high view) [its pred/loss discarded, Q12]
    -> MK_MMD(cf1, cf2) -> loss + loss_MDD -> argmax predictions -> backward -> optimizer.step()
The per-iteration host syncs of the reference (`.cpu()`, `.item()`, fusion_train.py:214-225) are
not part of the contract: metrics stay on the device and are pulled by the caller.
"""
import os

import torch

from . import ops
from .mmd import MK_MMD


def synthetic_batch(batch, H=224, W=224, S=32, device="cuda", seed=1234, rank=0, drop_oct_high=False):
    """Synthetic twin-view batch: low view ~U[0,1); high view = clip(low + N(0,0.5^2), 0, 1)
    (mirrors data_harvard.py:769-783); OCT-dropped high view = zeros (data_harvard.py:333-334)."""
    g = torch.Generator().manual_seed(seed + rank)
    f_low = torch.rand(batch, 3, H, W, generator=g)
    o_low = torch.rand(batch, 1, S, H, W, generator=g)
    f_high = (f_low + 0.5 * torch.randn(f_low.shape, generator=g)).clamp_(0, 1)
    if drop_oct_high:
        o_high = torch.zeros_like(o_low)
    else:
        o_high = (o_low + 0.5 * torch.randn(o_low.shape, generator=g)).clamp_(0, 1)
    y = torch.randint(0, 2, (batch,), generator=g, dtype=torch.int64)
    dev = torch.device(device)
    return ([f_low.to(dev), o_low.to(dev)], [f_high.to(dev), o_high.to(dev)]), y.to(dev)


def device_twin_views(fundus_low, oct_low, sigma=0.5, drop_oct_high=False):
    """make the high-noise view on the device from the resident low-noise view
    (Gaussian N(0, sigma^2) + clip, data_harvard.py:769-783; OCT-dropped = zeros, :333-334) instead of per-sample
    numpy on loader workers + a synchronous H2D copy (fusion_train.py:181-184)."""
    f_high = ops.twin_view(fundus_low, sigma)
    o_high = torch.zeros_like(oct_low) if drop_oct_high else ops.twin_view(oct_low, sigma)
    return [fundus_low, oct_low], [f_high, o_high]


# The two views' encoder passes run on two HIP streams (see train_step): the product default since round 5 -- identical losses,
# gradients and running statistics (tests/test_gpu_head.py::test_view_overlap_streams_same_results), +3 % (C1) / +7 % (C2).
# EDRL_VIEW_STREAM=0 / set_view_overlap(False): one view after the other on one stream (what per-kernel timing needs).
_VIEW_STREAM = os.environ.get("EDRL_VIEW_STREAM", "1") != "0"
_view_stream = None


def set_view_overlap(on):
    """Switch the two-stream execution of the two views' encoder passes (same results; see train_step)."""
    global _VIEW_STREAM
    _VIEW_STREAM = bool(on)
    if on:   # the shared parameters' AccumulateGrad nodes see gradients from two streams: intentional
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)


def view_overlap():
    return _VIEW_STREAM


if _VIEW_STREAM:
    set_view_overlap(True)


class DevicePrefetcher:
    """H2D side of SURVEY.md §8(f) row 3: wraps a loader of (X, y) batches with X = [fundus [B,3,H,W], oct [B,1,S,H,W]]
    host tensors, stages each batch through pinned buffers and copies it on a side stream one batch ahead of the
    consumer (the reference does a synchronous `.cuda()` per tensor, fusion_train.py:181-184), then builds the twin
    views on the device (device_twin_views: Gaussian + clip, optional OCT drop, optional salt-and-pepper).
    Yields ((low_views, high_views), y) ready for train_step."""

    def __init__(self, loader, device, sigma=0.5, drop_oct_high=False, salt_pepper=0.0):
        self.loader, self.device = loader, torch.device(device)
        self.sigma, self.drop, self.sp = sigma, drop_oct_high, salt_pepper
        self.stream = torch.cuda.Stream(self.device)
        self._pinned = {}

    def _stage(self, key, t):
        buf = self._pinned.get(key)
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = self._pinned[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        buf.copy_(t)
        return buf

    def _upload(self, batch, slot):
        X, y = batch
        with torch.cuda.stream(self.stream):
            dev = [self._stage((slot, i), t).to(self.device, non_blocking=True) for i, t in enumerate(X)]
            yd = self._stage((slot, "y"), y).to(self.device, non_blocking=True)
        return dev, yd

    def __iter__(self):
        it = iter(self.loader)
        nxt, slot = None, 0
        try:
            nxt = self._upload(next(it), slot)
        except StopIteration:
            return
        while nxt is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
            (fundus, oct_), y = nxt
            for t in (fundus, oct_, y):
                t.record_stream(torch.cuda.current_stream())
            slot ^= 1                       # the other pinned set: the copy that was just consumed may still be read
            try:
                nxt = self._upload(next(it), slot)
            except StopIteration:
                nxt = None
            low, high = device_twin_views(fundus, oct_, self.sigma, self.drop)
            if self.sp > 0:
                ops.salt_pepper_(high[0], self.sp)
                if not self.drop:
                    B, C, S, H, W = high[1].shape
                    ops.salt_pepper_(high[1].view(B * S, C, H, W), self.sp)
            yield (low, high), y


def train_step(model, optimizer, data, target, epoch=0, noise1=None, noise2=None, grad_sync=None):
    """One iteration of the loop body at fusion_train.py:176-225. Returns device tensors, no host sync.
    `grad_sync` (optional): a dist.GradSync (its zero_grad() keeps the gradients attached to the DP buckets, its finish()
    runs after backward and before optimizer.step), or a bare callable invoked after backward."""
    data1, data2 = data
    optimizer.zero_grad()
    if hasattr(grad_sync, "zero_grad"):
        grad_sync.zero_grad()          # re-attaches the bucket views as .grad (zeroed) after the set_to_none above
    trunks = model.trunks() if (hasattr(model, "trunks") and model.training and target.is_cuda) else ()
    for t in trunks:                   # weight shadows for both views; the views' parameter gradients are summed per stage
        t.step_begin()
    ops.step_cache_begin()             # (the head's permuted Linear weights: one copy per step instead of one per view)
    try:
        return _train_step_body(model, optimizer, data1, data2, target, epoch, noise1, noise2, grad_sync)
    finally:
        ops.step_cache_end()
        for t in trunks:
            t.step_end()


def _train_step_body(model, optimizer, data1, data2, target, epoch, noise1, noise2, grad_sync):
    if _VIEW_STREAM and target.is_cuda and model.training:
        # The two views' encoder passes are independent (they meet in MK_MMD): the second one runs on a side stream so
        # that its HBM-bound BatchNorm kernels overlap the first one's MFMA-bound convolutions and vice versa; autograd
        # replays the same streams in backward.  The head (stateful DILR.bn1/bn2, 4 updates per step) stays in order
        # on the main stream; the encoders' running statistics are merged in order (ResNetTrunk.merge_scratch_running).
        global _view_stream
        main = torch.cuda.current_stream()
        if _view_stream is None:
            _view_stream = torch.cuda.Stream()
        side = _view_stream
        side.wait_stream(main)
        model.check_labels(target)
        tok1 = model.encode(data1)
        with torch.cuda.stream(side):
            for t in model.trunks():
                t.begin_scratch_running()
            tok2 = model.encode(data2)
            for t in model.trunks():
                t.end_scratch_running()
        pred, loss, combined_features1 = model.forward_tokens(tok1[0], tok1[1], target, noise1)
        main.wait_stream(side)
        for t in model.trunks():
            t.merge_scratch_running()
        for t in tok2:
            t.record_stream(main)
        _, _, combined_features2 = model.forward_tokens(tok2[0], tok2[1], target, noise2)
    else:
        pred, loss, combined_features1 = model(data1, target, epoch, noise=noise1)
        _, _, combined_features2 = model(data2, target, epoch, noise=noise2)
    loss_MDD = MK_MMD(combined_features1, combined_features2)
    total = ops.scalar_mix([1.0, 1.0], [loss, loss_MDD])
    predicted = ops.argmax_rows(pred)
    mark = getattr(grad_sync, "mark_backward", None)      # GradSync diagnostics (no-op unless enabled)
    if mark is not None:
        mark(True)
    total.backward()
    if mark is not None:
        mark(False)
    if grad_sync is not None:
        (grad_sync.finish if hasattr(grad_sync, "finish") else grad_sync)()
    optimizer.step()
    return {"loss": total.detach(), "loss_MDD": loss_MDD.detach(), "pred": pred.detach(), "predicted": predicted}


def train(epoch, train_loader, model, optimizer, log_every=0):
    """fusion_train.train(): iterate the loader, accumulate accuracy/loss (pulled once at the end)."""
    model.train()
    correct = torch.zeros((), device=next(model.parameters()).device, dtype=torch.int64)
    loss_sum = torch.zeros((), device=correct.device)
    n_batches, n_seen = 0, 0
    for data, target in train_loader:
        out = train_step(model, optimizer, data, target, epoch)
        correct += (out["predicted"] == target).sum()
        loss_sum += out["loss"]
        n_batches += 1
        n_seen += target.shape[0]
    if hasattr(model, "raise_on_bad_labels"):
        model.raise_on_bad_labels()
    if hasattr(model, "raise_on_nonfinite"):
        model.raise_on_nonfinite()
    return {"loss": (loss_sum / max(n_batches, 1)).item(), "acc": correct.item() / max(n_seen, 1)}


@torch.no_grad()
def val(current_epoch, val_loader, model, best_acc, save_path=None):
    """fusion_train.val() (fusion_train.py:267-334): eval-mode forward on the low-noise view only (:277),
    accuracy / mean loss, and the best-accuracy checkpoint `{'epoch', 'state_dict'}` (:324-332)."""
    model.eval()
    dev = next(model.parameters()).device
    correct = torch.zeros((), device=dev, dtype=torch.int64)
    loss_sum = torch.zeros((), device=dev)
    n_batches, n_seen = 0, 0
    for data, target in val_loader:
        pred, loss, _ = model(data[0], target, current_epoch)
        correct += (ops.argmax_rows(pred) == target).sum()
        loss_sum += loss
        n_batches += 1
        n_seen += target.shape[0]
    if hasattr(model, "raise_on_bad_labels"):
        model.raise_on_bad_labels()
    acc = correct.item() / max(n_seen, 1)
    if acc > best_acc and save_path is not None:
        torch.save({"epoch": current_epoch, "state_dict": model.state_dict()}, save_path)
    return {"loss": (loss_sum / max(n_batches, 1)).item(), "acc": acc, "best_acc": max(acc, best_acc)}
