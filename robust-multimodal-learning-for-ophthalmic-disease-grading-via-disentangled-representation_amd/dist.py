"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm)
all-reduce of flat fp32 gradient buckets over xGMI, issued on a side stream as soon as a bucket's
gradients exist so that it overlaps the rest of backward.

The reference has no distributed code (one dead all_reduce, fusion_net.py:686); this is the build's
own DP layer.  Batch-coupled statistics (BatchNorm, bt_loss_cross, MK_MMD) stay per replica — the
semantics DistributedDataParallel would give the reference — and `args.batch_size` is the per-GPU batch.

Buckets are filled in reverse registration order (the order backward produces gradients); parameters
that never receive a gradient (dead modules; EPRL.alpha/decoder_logits/mlp_*) are left out after the
first step.  The exchange is a SUM followed by a 1/world scale, in place on the flat bucket.
"""
import torch
import torch.distributed as dist


class GradSync:
    def __init__(self, model, bucket_mb=64, process_group=None):
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in model.parameters() if p.requires_grad][::-1]
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.buckets = None          # list of dicts {params, flat, ready, handle}
        self.index = {}              # param -> (bucket id)
        self.comm_stream = None
        self._hooks = []
        self._step_active = False

    # -- first step: no hooks yet; learn which parameters receive gradients, then build buckets
    def _build(self):
        live = [p for p in self.params if p.grad is not None]
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in live:
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= self.bucket_bytes:
                self.buckets.append(cur); cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        built = []
        for bi, ps in enumerate(self.buckets):
            n = sum(p.numel() for p in ps)
            flat = torch.empty(n, device=ps[0].device, dtype=torch.float32)
            views, off = [], 0
            for p in ps:
                views.append(flat[off:off + p.numel()].view_as(p)); off += p.numel()
                self.index[p] = bi
            built.append({"params": ps, "flat": flat, "views": views, "ready": 0, "handle": None, "event": None})
        self.buckets = built
        if live and live[0].is_cuda:
            self.comm_stream = torch.cuda.Stream()
        for p in live:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _on_grad(self, p):
        b = self.buckets[self.index[p]]
        b["ready"] += 1
        if b["ready"] == len(b["params"]):
            self._launch(b)

    def _launch(self, b):
        grads = [p.grad for p in b["params"]]
        if self.comm_stream is not None:
            ev = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                torch._foreach_copy_(b["views"], grads)
                if self.world > 1:
                    b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        else:
            torch._foreach_copy_(b["views"], grads)
            if self.world > 1:
                b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def finish(self):
        """Call after backward and before optimizer.step(): averages every gradient over the ranks."""
        if self.buckets is None:
            self._build()
            for b in self.buckets:      # first step: nothing was launched from hooks
                b["ready"] = len(b["params"])
                self._launch(b)
        inv = 1.0 / self.world
        for b in self.buckets:
            if b["ready"] != len(b["params"]):
                raise RuntimeError("GradSync.finish(): a bucket is incomplete (a parameter stopped receiving gradients)")
            if b["handle"] is not None:
                b["handle"].wait()       # makes the current stream wait for the collective
                b["handle"] = None
            if self.comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
            if self.world > 1:
                b["flat"].mul_(inv)
            torch._foreach_copy_([p.grad for p in b["params"]], b["views"])
            b["ready"] = 0

    def total_bytes(self):
        return sum(b["flat"].numel() * 4 for b in (self.buckets or []))


def broadcast_parameters(model, src=0, group=None):
    """Identical initial weights on every rank (parameters and buffers)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
