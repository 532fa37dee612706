"""CPU, world_size 2, gloo: the DP gradient exchange (GradSync) — bucketing, hook-driven launches after the
first step on, gradients living in the flat buckets, dead parameters skipped, two forwards sharing weights, a detached-view
(set_to_none) step, a gradient-accumulation (no_sync) step, and N-rank averaged gradients equal to the single-process
gradients on the concatenated batch for batch-decoupled losses."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(16, 32)
        self.b = nn.Linear(32, 8)
        self.dead = nn.Linear(4, 4)          # never used: must stay out of the buckets
        self.c = nn.Linear(8, 1)

    def forward(self, x):
        return self.c(torch.relu(self.b(torch.relu(self.a(x)))))


def _loss(net, x1, x2):
    # two forwards share the weights (fusion_train.py:191,194); per-sample mean -> batch-decoupled
    return net(x1).pow(2).mean() + 0.5 * net(x2).abs().mean()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import edrl_amd
    torch.manual_seed(0)
    net = Net()
    edrl_amd.broadcast_parameters(net)
    live = [p for n, p in net.named_parameters() if not n.startswith("dead")]   # (MedFusion supplies live_parameters() itself)
    sync = edrl_amd.GradSync(net, bucket_mb=0.001, params=live)     # tiny buckets -> several collectives
    views = {n: p.grad.data_ptr() for n, p in net.named_parameters() if p.grad is not None}
    g = torch.Generator().manual_seed(5)
    X1, X2 = torch.randn(4, 8, 16, generator=g), torch.randn(4, 8, 16, generator=g)   # 4 steps, global batch 8
    per = 8 // world
    res = []
    for step in range(4):
        if step == 2:
            net.zero_grad()          # set_to_none=True detaches the bucket views: the hooks must put the gradients back
        else:
            sync.zero_grad()
        x1, x2 = X1[step, rank * per:(rank + 1) * per], X2[step, rank * per:(rank + 1) * per]
        if step == 3:                # gradient accumulation: two half micro-batches, exchange only once
            h = per // 2
            with sync.no_sync():
                (_loss(net, x1[:h], x2[:h]) * 0.5).backward()
            (_loss(net, x1[h:], x2[h:]) * 0.5).backward()
        elif step == 0:              # exchange diagnostics (bench.py's N > 1 JSON fields), host-clock part: no HIP events on the CPU
            sync.enable_diagnostics(True)
            sync.mark_backward(True)
            _loss(net, x1, x2).backward()
            sync.mark_backward(False)
        else:
            _loss(net, x1, x2).backward()
        sync.finish()
        if step == 0:
            rep = sync.step_report()
            sync.enable_diagnostics(False)
        res.append({n: p.grad.numpy().copy() for n, p in net.named_parameters() if p.grad is not None})
        assert all(p.grad.data_ptr() == views[n] for n, p in net.named_parameters() if p.grad is not None), \
            "gradients must live in the flat buckets (no gather/scatter copies)"
    q.put((rank, res, len(sync.buckets), sync.total_bytes(), rep))
    dist.destroy_process_group()


def test_gradsync_world2_matches_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    outs.sort(key=lambda o: o[0])
    torch.manual_seed(0)
    net = Net()
    g = torch.Generator().manual_seed(5)
    X1, X2 = torch.randn(4, 8, 16, generator=g), torch.randn(4, 8, 16, generator=g)
    live_bytes = sum(p.numel() * 4 for n, p in net.named_parameters() if not n.startswith("dead"))
    for rank, res, nb, nbytes, rep in outs:
        assert nb > 1, "expected several buckets"
        # every bucket exchanged once, the first one issued by a hook while backward was still running, bytes = live parameters
        assert rep["collectives_this_step"] == nb and rep["buckets"] == nb and rep["bytes_exchanged"] == live_bytes, rep
        assert rep["first_launch_host_ms_before_backward_end"] > 0 and not rep["launched_in_finish"][0], rep
        assert nbytes == live_bytes, "dead parameters must not be exchanged"
        for step in range(4):
            net.zero_grad()
            _loss(net, X1[step], X2[step]).backward()
            assert "dead.weight" not in res[step]
            for n, p in net.named_parameters():
                if p.grad is None:
                    continue
                torch.testing.assert_close(torch.from_numpy(res[step][n]), p.grad, rtol=1e-5, atol=1e-6)


def test_gradsync_rejects_second_backward_outside_no_sync():
    """a second backward() between zero_grad() and finish() outside no_sync() would add gradients on top of an
    already exchanged (averaged) bucket and the ranks would diverge silently; GradSync raises instead.  Inside no_sync() the
    same pattern is gradient accumulation and is exchanged once at finish().  (No process group: single replica, CPU.)"""
    import edrl_amd
    torch.manual_seed(0)
    net = Net()
    live = [p for n, p in net.named_parameters() if not n.startswith("dead")]
    sync = edrl_amd.GradSync(net, bucket_mb=0.001, params=live)
    x = torch.randn(8, 16)
    sync.zero_grad()
    net(x).sum().backward()
    with pytest.raises(RuntimeError, match="already exchanged"):
        net(x).sum().backward()
    sync.finish()
    # accumulation the supported way: both backward passes land in the buckets, one exchange
    sync.zero_grad()
    with sync.no_sync():
        net(x).sum().backward()
    net(x).sum().backward()
    sync.finish()
    ref = Net()
    ref.load_state_dict(net.state_dict())
    (ref(x).sum() * 2).backward()
    for (n, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
        if r.grad is not None and not n.startswith("dead"):
            torch.testing.assert_close(p.grad, r.grad, rtol=1e-5, atol=1e-6)
    assert sync.collectives_issued == 0            # one replica: nothing to exchange unless force_collective
