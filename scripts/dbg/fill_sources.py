import sys, os, types, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, edrl_amd
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda:0")
args = types.SimpleNamespace(mode="train", batch_size=2, encoder_depth=50)
torch.manual_seed(0)
m = edrl_amd.MedFusion(2, 2, None, args).to(dev).train()
opt = edrl_amd.FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-6)
data, y = edrl_amd.synthetic_batch(2, 64, 64, 4, device=dev)
for _ in range(2):
    edrl_amd.train_step(m, opt, data, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
    edrl_amd.train_step(m, opt, data, y)
    torch.cuda.synchronize()
cnt = collections.Counter()
def chain(e):
    out = []
    p = e.cpu_parent
    while p is not None and len(out) < 3:
        out.append(p.name[:40]); p = p.cpu_parent
    return " < ".join(out)
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::copy_", "aten::clone"):
        shp = str(e.input_shapes[0]) if e.input_shapes else "?"
        nd = len(e.input_shapes[0]) if e.input_shapes and e.input_shapes[0] else 0
        cnt[(e.name, chain(e), nd)] += 1
for (n, c, nd), k in cnt.most_common(30):
    print(f"{k:5d}  {n:12s} ndim={nd}  parents: {c}")
