"""Diagnostic (GPU): the C0 full step vs the fp64 oracle -- where does the head-gradient error come from?"""
import os, sys, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import edrl_amd as edrl
from oracle import step_oracle as SO, resnet_oracle as RO
depth, B, HW, S = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (18, 2, 224, 16))]
dev = torch.device("cuda:0")
args = types.SimpleNamespace(mode="train", batch_size=B, encoder_depth=depth)
torch.manual_seed(0)
m = edrl.MedFusion(2, 2, None, args).to(dev).train()
orc = SO.OracleEDRL(m, dtype=torch.float64)
data, y = edrl.synthetic_batch(B, HW, HW, S, device="cpu")
N2, N3 = (HW // 32) ** 2, S
n1, n2 = SO.make_noise(50, B, N2, N3), SO.make_noise(51, B, N2, N3)
to_dev = lambda o: {k: to_dev(v) for k, v in o.items()} if isinstance(o, dict) else o.to(dev)
cast = lambda o: {k: cast(v) for k, v in o.items()} if isinstance(o, dict) else o.double()
opt = torch.optim.Adam(m.parameters(), lr=1e-4, weight_decay=1e-6)
sels = []
orig = edrl.ops.topk_margin
def rec(att, yy, K=100):
    l, s = orig(att, yy, K)
    sels.append((att.detach().cpu(), s.cpu()))
    return l, s
edrl.ops.topk_margin = rec
import importlib
mf = sys.modules[edrl.MedFusion.__module__]
mf.ops.topk_margin = rec
seqs = {k: [] for k in ("transformer_2DNet", "transformer_3DNet")}
for k, sq in seqs.items():
    getattr(m, k).trunk._capture_seq = sq
ddev = ([t.to(dev) for t in data[0]], [t.to(dev) for t in data[1]])
out = edrl.train_step(m, opt, ddev, y.to(dev), noise1=to_dev(n1), noise2=to_dev(n2))
pins = [(RO.pins_from_capture(seqs["transformer_2DNet"][v]), RO.pins_from_capture(seqs["transformer_3DNet"][v])) for v in (0, 1)]
# oracle forward per view to get aux (selection indices)
params = orc.parameters()
pred, loss, cf1, aux1 = orc.forward([data[0][0].double(), data[0][1].double()], y, cast(n1), pins[0])
_, _, cf2, aux2 = orc.forward([data[1][0].double(), data[1][1].double()], y, cast(n2), pins[1])
from oracle import edrl_oracle as O
total = loss + O.MK_MMD(cf1, cf2)
total.backward()
print("n product topk calls:", len(sels))
for vi, aux in enumerate((aux1, aux2)):
    for mi, key in enumerate(("sel_fundus", "sel_oct")):
        att_p, sel_p = sels[vi * 2 + mi]
        a = aux[key]
        att_o = a["att"]
        ea = ((att_p.double() - att_o).abs().max() / att_o.abs().max()).item()
        # oracle selection -> mask
        Bn, C, Sn = att_o.shape
        ref = torch.zeros(Bn, C, Sn, dtype=torch.bool)
        for b in range(Bn):
            ref[b, int(y[b]), a["idx_pos"][b]] = True
            ref[b, 1 - int(y[b]), a["idx_neg"][b]] = True
        ndiff = int((ref != sel_p.bool()).sum())
        # gap at the selection boundary
        gaps = []
        for b in range(Bn):
            for c in range(C):
                v = att_o[b, c].sort(descending=True).values
                gaps.append(((v[99] - v[100]) / v.abs().max()).item())
        print(f"view {vi} {key}: att rel err {ea:.3e}; selection entries differing {ndiff}; min rel gap rank100/101 {min(gaps):.3e}")
named = dict(m.named_parameters())
rows = []
for n, t in params.items():
    key = n
    if ".trunk." in n:
        h, tl = n.split(".trunk."); key = h + ".trunk." + tl.replace(".", "__")
    g = named[key].grad.cpu().double(); r = t.grad
    sc = r.abs().max().clamp_min(1e-12)
    d = (g - r).abs()
    rows.append((float(d.max() / sc), n, float((d > 1e-4 * sc).double().mean()), float((g - r).norm() / r.norm().clamp_min(1e-30))))
rows.sort(reverse=True)
for e, n, frac, fro in rows[:25]:
    print(f"{e:.3e}  fro {fro:.3e}  frac>1e-4 {frac:.3f}  {n}")
