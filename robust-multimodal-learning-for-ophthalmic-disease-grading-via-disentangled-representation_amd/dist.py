"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl" on ROCm)
all-reduce of flat fp32 gradient buckets over xGMI, issued on a side stream as soon as a bucket's
gradients exist so that it overlaps the rest of backward.

The reference has no distributed code (one dead all_reduce, fusion_net.py:686); this is the build's
own DP layer.  Batch-coupled statistics (BatchNorm, bt_loss_cross, MK_MMD) stay per replica — the
semantics DistributedDataParallel would give the reference — and `args.batch_size` is the per-GPU batch.

Layout.  The `.grad` of every exchanged parameter IS a view of its bucket's flat buffer (no gather / scatter
copies): autograd accumulates into the bucket in place, the collective runs on the bucket in place (SUM, then
x 1/world on the communication stream), and the optimiser reads the averaged gradients where they lie.  Buckets
follow reverse registration order (the order backward produces gradients).  `GradSync.zero_grad()` replaces
`optimizer.zero_grad()`: one fill per bucket, the views stay attached.

Which parameters are exchanged is fixed at construction: `model.live_parameters()` when the model provides it
(MedFusion does: its dead modules / eval-only parameters never receive gradients), otherwise every
parameter that requires grad.  Hooks are armed from the very first step, so the first step overlaps like every other.
A bucket whose parameters did not all report by `finish()` (a parameter unused in this step) is exchanged there with
the zeros `zero_grad()` left in it.
"""
import contextlib
import datetime
import os
import time

import torch
import torch.distributed as dist


def init_process_group(backend="nccl", device=None, timeout_s=180):
    """torch.distributed initialisation with fail-fast error handling: a failed or hung RCCL collective aborts the
    process (after `timeout_s`) instead of leaving the other ranks blocked in backward."""
    os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")      # tear the process down on a collective error
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC (the only mode this host driver supports)
    try:                                                                # (see the package __init__: the two compute streams need queues of their own)
        if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
            import warnings
            warnings.warn("GPU_MAX_HW_QUEUES < 8: with a process group active the two-view stream overlap of train_step shares a "
                          "hardware queue with the communication streams (-4 % step throughput at C1); export GPU_MAX_HW_QUEUES=8 "
                          "before the process touches the GPU")
    except ValueError:
        pass
    kw = dict(timeout=datetime.timedelta(seconds=timeout_s))
    if backend == "nccl" and device is not None:
        kw["device_id"] = device
    dist.init_process_group(backend, **kw)


class GradSync:
    def __init__(self, model, bucket_mb=64, process_group=None, params=None, force_collective=False):
        """`force_collective`: with a single rank the exchange is the identity and is normally skipped; True issues the
        all-reduce / wait / scale on the communication stream anyway (needs an initialised process group), so that the
        RCCL code path of the N > 1 run can be executed and checked on a one-GPU box (tests/test_gpu_dist.py)."""
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.force_collective = bool(force_collective) and dist.is_initialized()
        self.collectives_issued = 0      # all-reduces handed to the backend since construction (diagnostics / tests)
        # step diagnostics (enable_diagnostics): host timestamps + HIP events of backward start / end, of every bucket launch and
        # around finish()'s wait on the communication stream -- what a first multi-GPU run needs to be read from its JSON
        self._diag = None
        if params is None:
            params = model.live_parameters() if hasattr(model, "live_parameters") else model.parameters()
        self.params = [p for p in params if p.requires_grad][::-1]
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.index = {}              # param -> bucket id
        self.slot = {}               # param -> position inside its bucket
        self.comm_stream = None
        self._sync = True
        self._hooks = []
        groups, cur, cur_bytes = [], [], 0
        for p in self.params:
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= self.bucket_bytes:
                groups.append(cur); cur, cur_bytes = [], 0
        if cur:
            groups.append(cur)
        self.buckets = []
        for bi, ps in enumerate(groups):
            n = sum(p.numel() for p in ps)
            flat = torch.zeros(n, device=ps[0].device, dtype=torch.float32)
            views, off = [], 0
            for p in ps:
                v = flat[off:off + p.numel()].view_as(p)
                views.append(v); off += p.numel()
                self.index[p] = bi
                self.slot[p] = len(views) - 1
                p.grad = v                                   # autograd accumulates into the bucket in place
            self.buckets.append({"params": ps, "flat": flat, "views": views, "ready": 0, "launched": False, "handle": None,
                                 "seen": [False] * len(ps)})
        # share of each bucket's elements that are encoder-trunk parameters (diagnostics: which launches carry encoder gradients)
        names = {p: n for n, p in model.named_parameters()} if hasattr(model, "named_parameters") else {}
        for bi, b in enumerate(self.buckets):
            b["index"] = bi
            enc = sum(p.numel() for p in b["params"] if ".trunk." in names.get(p, ""))
            b["encoder_fraction"] = enc / max(b["flat"].numel(), 1)
        if self.params and self.params[0].is_cuda:
            self.comm_stream = torch.cuda.Stream()
        # The encoder trunks sum the two views' parameter gradients themselves, stage by stage, straight into the bucket views
        # (encoders._TrunkFn.backward inside a train_step) and report every finished residual stage to params_ready -- so encoder
        # buckets go out during backward, not after the last trunk node has returned.  The trunk nodes then hand the engine no
        # gradient for those parameters; their post-accumulate hooks still fire (with nothing accumulated) once the last node
        # has returned and are ignored while the trunk manages its gradients itself (ResNetTrunk.stash_active).
        self._trunk_of = {}
        for t in (model.trunks() if hasattr(model, "trunks") else ()):
            t.grad_sink = self.params_ready
            for p in t.parameters():
                self._trunk_of[p] = t
        self.hook_calls_ignored = 0
        for p in self.params:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._hook))

    # ---- step protocol: zero_grad() -> forward/backward (hooks launch full buckets) -> finish() -> optimizer.step()
    def zero_grad(self):
        """Zero every exchanged gradient (one fill per bucket; the `.grad` views stay attached) and re-arm the step."""
        for b in self.buckets:
            if b["handle"] is not None:                      # a step abandoned between backward and finish()
                b["handle"].wait(); b["handle"] = None
            b["flat"].zero_()
            b["ready"], b["launched"] = 0, False
            b["seen"] = [False] * len(b["params"])
            for p, v in zip(b["params"], b["views"]):
                if p.grad is not v:                          # someone ran optimizer.zero_grad(set_to_none=True)
                    p.grad = v

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation: backward passes inside this context only accumulate into the buckets; the exchange
        happens in the first finish() outside it."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def _hook(self, p):
        t = self._trunk_of.get(p)
        if t is not None and t.stash_active():
            self.hook_calls_ignored += 1
            return
        self._on_grad(p)

    def _on_grad(self, p):
        b = self.buckets[self.index[p]]
        v = b["views"][self.slot[p]]
        if p.grad is not None and p.grad.data_ptr() != v.data_ptr():
            v.copy_(p.grad)                                  # the view was detached (set_to_none): put the gradient back
            p.grad = v
        if not self._sync:
            return
        # One backward per zero_grad()/finish() outside no_sync(): a second gradient for a parameter of an exchanged
        # bucket would be added on top of an already averaged buffer and the ranks would diverge silently.
        if b["launched"]:
            raise RuntimeError("GradSync: a gradient arrived for a bucket that was already exchanged in this step; run "
                               "extra backward passes inside no_sync() and call finish() once after the last one")
        si = self.slot[p]
        if not b["seen"][si]:
            b["seen"][si] = True
            b["ready"] += 1
        if self.comm_stream is not None:                     # streams whose work this bucket's gradients come from
            cur = torch.cuda.current_stream()
            b.setdefault("streams", {})[cur.cuda_stream] = cur
        if b["ready"] >= len(b["params"]):
            self._launch(b)

    def params_ready(self, params):
        """The `.grad` of these parameters is complete for this step (the encoder trunks call this per residual stage)."""
        for p in params:
            if p in self.index:
                self._on_grad(p)

    # ---- diagnostics ------------------------------------------------------------------------------------------------------
    def enable_diagnostics(self, on=True):
        """Record, per step: when backward started / ended and when each bucket's all-reduce was ISSUED (host clock and an
        event on the compute stream), and how long finish() had the compute stream wait for the communication stream
        (`comm_exposed_ms`: the part of the exchange that backward did not hide).  Read with step_report() after a
        synchronize.  Costs a handful of events per step; off by default."""
        self._diag = {"launch": [], "bwd": [None, None], "wait": None} if on else None

    def mark_backward(self, start):
        """train_step calls this right before / after `loss.backward()` (no-op unless diagnostics are enabled)."""
        d = self._diag
        if d is None:
            return
        if start:
            d["launch"], d["wait"] = [], None
        ev = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True)) if self.comm_stream is not None else None
        d["bwd"][0 if start else 1] = (time.perf_counter(), ev)

    def step_report(self):
        """Diagnostics of the last finished step (call after torch.cuda.synchronize()): dict, or None when disabled."""
        d = self._diag
        if d is None or d["bwd"][0] is None or d["bwd"][1] is None:
            return None
        (t0, e0), (t1, e1) = d["bwd"]
        rep = {"buckets": len(self.buckets), "bytes_exchanged": self.total_bytes(), "collectives_this_step": len(d["launch"]),
               "backward_host_ms": round((t1 - t0) * 1e3, 3),
               "launch_host_ms_after_backward_start": [round((t - t0) * 1e3, 3) for t, _, _, _ in d["launch"]],
               "launched_in_finish": [bool(f) for _, _, f, _ in d["launch"]],
               "launch_bucket_index": [bi for _, _, _, bi in d["launch"]],
               "launch_encoder_fraction": [round(self.buckets[bi]["encoder_fraction"], 3) for _, _, _, bi in d["launch"]]}
        if d["launch"]:
            rep["first_launch_host_ms_before_backward_end"] = round((t1 - d["launch"][0][0]) * 1e3, 3)
        if e0 is not None and e1 is not None:
            rep["backward_gpu_ms"] = round(e0.elapsed_time(e1), 3)
            rep["launch_gpu_ms_after_backward_start"] = [round(e0.elapsed_time(ev), 3) for _, ev, _, _ in d["launch"] if ev is not None]
            if d["wait"] is not None:
                rep["comm_exposed_ms"] = round(d["wait"][0].elapsed_time(d["wait"][1]), 3)
        return rep

    def _launch(self, b, in_finish=False):
        b["launched"] = True
        if self._diag is not None and (self.world > 1 or self.force_collective):
            ev = torch.cuda.current_stream().record_event(torch.cuda.Event(enable_timing=True)) if self.comm_stream is not None else None
            self._diag["launch"].append((time.perf_counter(), ev, in_finish, b.get("index", -1)))
        if self.world == 1 and not self.force_collective:
            return
        inv = 1.0 / self.world
        self.collectives_issued += 1
        if self.comm_stream is not None:
            # The exchange waits for every stream that produced one of the bucket's gradients (the two views run on two streams:
            # normally the trunks' last pass and the head are all on one of them and has already waited for the other's stage
            # events, but nothing forces a bucket's parameters onto one stream).
            cur = torch.cuda.current_stream()
            evs = [cur.record_event()] + [st.record_event() for sid, st in b.get("streams", {}).items() if sid != cur.cuda_stream]
            b["streams"] = {}
            with torch.cuda.stream(self.comm_stream):
                for ev in evs:
                    self.comm_stream.wait_event(ev)
                b["handle"] = dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                b["handle"].wait()                           # orders the scale after the collective ON the comm stream
                b["handle"] = None
                b["flat"].mul_(inv)
        else:
            dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.group)
            b["flat"].mul_(inv)

    def finish(self):
        """Call after backward and before optimizer.step(): every gradient is the average over the ranks afterwards."""
        for b in self.buckets:
            if not b["launched"]:                            # a parameter of this bucket saw no gradient in this step
                self._launch(b, in_finish=True)
        if self.comm_stream is not None:
            cur = torch.cuda.current_stream()
            w0 = cur.record_event(torch.cuda.Event(enable_timing=True)) if self._diag is not None else None
            cur.wait_stream(self.comm_stream)
            if w0 is not None:
                self._diag["wait"] = (w0, cur.record_event(torch.cuda.Event(enable_timing=True)))
        for b in self.buckets:                               # re-arm: the next backward starts a new exchange even if the
            b["ready"], b["launched"] = 0, False             # caller zeroes gradients some other way than zero_grad()
            b["seen"] = [False] * len(b["params"])

    def total_bytes(self):
        return sum(b["flat"].numel() * 4 for b in self.buckets)


def broadcast_parameters(model, src=0, group=None, force_collective=False):
    """Identical initial weights on every rank (parameters and buffers).  `force_collective` issues the broadcasts with a
    single rank too (the identity; exercises the backend on a one-GPU box)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective):
        return
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, src=src, group=group)
