"""FusedAdam: drop-in for `optim.Adam(model.parameters(), lr=args.lr, weight_decay=1e-6)` (fusion_train.py:747) whose
`step()` is ONE launch of the multi-tensor HIP kernel `edrl_adam_multi_f32`.

Same update rule as torch.optim.Adam (betas, eps, L2 weight decay folded into the gradient, no amsgrad / maximize);
the per-parameter state uses torch's keys (`step`, `exp_avg`, `exp_avg_sq`), so `state_dict()` / `load_state_dict()`
interchange with torch.optim.Adam checkpoints.  Parameters without a gradient are skipped, exactly like torch.
"""
import struct

import torch

from . import _lib as L


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._chunk_cache = {}     # (sizes tuple) -> device chunk table
        self._table_cache = {}     # number of tensors -> (pointer records, device copy)

    def _chunks(self, sizes, device):
        key = (tuple(sizes), str(device))
        hit = self._chunk_cache.get(key)
        if hit is None:
            ce = L.lib().fn["edrl_adam_chunk_elems"]()
            recs = []
            for ti, n in enumerate(sizes):
                recs.extend((ti, c) for c in range((n + ce - 1) // ce))
            tab = torch.tensor(recs, dtype=torch.int32).reshape(-1, 2).to(device)
            hit = self._chunk_cache[key] = (tab, len(recs))
        return hit

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.dtype != torch.float32 or not p.is_cuda or p.grad.is_sparse:
                    raise RuntimeError("FusedAdam: fp32 dense parameters on the GPU only (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                by_step.setdefault(int(st["step"].item()), []).append(p)
            for step, ps in by_step.items():      # one launch per distinct step count (one, in practice)
                recs, sizes, keep = [], [], []
                for p in ps:
                    st = self.state[p]
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    pc = p.data
                    if not (pc.is_contiguous() and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                        raise RuntimeError("FusedAdam: non-contiguous parameter/state")
                    keep.append(g)
                    recs.append(struct.pack("QQQQq", pc.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(),
                                            st["exp_avg_sq"].data_ptr(), p.numel()))
                    sizes.append(p.numel())
                dev = ps[0].device
                blob = b"".join(recs)
                cached = self._table_cache.get(step_key := len(ps))
                if cached is not None and cached[0] == blob:
                    table = cached[1]       # steady state: the caching allocator hands the gradients the same blocks
                else:                       # (a pageable H2D copy synchronises the host with the stream: only on change)
                    table = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
                    self._table_cache[step_key] = (blob, table)
                chunks, n_chunks = self._chunks(sizes, dev)
                b1, b2 = group["betas"]
                L.call("edrl_adam_multi_f32", table.data_ptr(), len(ps), chunks.data_ptr(), n_chunks, float(group["lr"]),
                       float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]), step)
                del keep
        return loss
