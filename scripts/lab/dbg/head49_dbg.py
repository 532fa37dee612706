"""Diagnostic (GPU): head-only step at odd token counts vs the fp64 oracle."""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import edrl_amd as edrl
from oracle import edrl_oracle as O
dev = torch.device("cuda:0")
to_dev = lambda o: {k: to_dev(v) for k, v in o.items()} if isinstance(o, dict) else o.to(dev)
cast = lambda o: {k: cast(v) for k, v in o.items()} if isinstance(o, dict) else o.double()
for (B, N2, N3) in ((2, 49, 16), (2, 48, 16), (2, 9, 6), (8, 49, 16), (2, 50, 16)):
    args = types.SimpleNamespace(mode="train", batch_size=B, encoder_depth=18)
    torch.manual_seed(0)
    m = edrl.MedFusion(2, 2, None, args)
    m.load_state_dict(O.make_head_params(5), strict=False)
    m = m.to(dev).train()
    x, x1, y, noise = O.make_head_inputs(6, B, N2, N3)
    p = {n: t.detach().cpu().double().requires_grad_(True) for n, t in m.named_parameters() if n in O.head_param_shapes()}
    st = O.make_bn_state(torch.float64)
    pred_o, loss_o, cf_o, aux = O.medfusion_forward_tokens(p, st, x.double(), x1.double(), y, cast(noise), B)
    loss_o.backward()
    pred, loss, cf = m.forward_tokens(x.to(dev), x1.to(dev), y.to(dev), to_dev(noise))
    loss.backward()
    rows = []
    for n, t in p.items():
        g = dict(m.named_parameters())[n].grad
        if t.grad is None or g is None: continue
        g = g.cpu().double(); r = t.grad
        sc = r.abs().max().clamp_min(1e-30); d = (g - r).abs()
        idx = int(d.argmax())
        rows.append((float(d.max() / sc), n, tuple(int(v) for v in torch.unravel_index(torch.tensor(idx), r.shape)), float((d > 1e-4 * sc).double().mean())))
    rows.sort(reverse=True)
    print(f"B={B} N2={N2} N3={N3}: pred err {float((pred.cpu().double()-pred_o).abs().max()/pred_o.abs().max()):.2e}")
    for e, n, ix, fr in rows[:5]:
        print(f"   {e:.3e} at {ix} frac {fr:.4f} {n}")
