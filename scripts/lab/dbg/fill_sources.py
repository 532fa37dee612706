"""Where do the small aten fill / add / copy kernels of a training step come from?  One C1-shaped step at a reduced batch under
torch.profiler with Python stacks; aten::fill_ / zero_ / add_ / copy_ / zeros calls aggregated by the innermost frame inside the
package (diagnostic, GPU).  usage: python scripts/dbg/fill_sources.py"""
import os, sys, collections, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import edrl_amd
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda:0")
B = 4
args = types.SimpleNamespace(mode="train", batch_size=B, encoder_depth=50, activation_recompute=False, encoder_dtype="fp32",
                             oct_encoder="slices", oct3d_depth=18)
torch.manual_seed(0)
model = edrl_amd.MedFusion(2, 2, None, args).to(dev).train()
opt = edrl_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-6)
data, y = edrl_amd.synthetic_batch(B, 224, 224, 32, device=dev, seed=1234, rank=0)
for _ in range(2):
    edrl_amd.train_step(model, opt, data, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    edrl_amd.train_step(model, opt, data, y)
    torch.cuda.synchronize()
acc = collections.Counter()
want = ("aten::fill_", "aten::zero_", "aten::add_", "aten::add", "aten::copy_", "aten::zeros", "aten::zeros_like", "aten::clone", "aten::contiguous")
for ev in prof.events():
    if ev.name in want:
        st = [s for s in (ev.stack or []) if "_amd/" in s or "bench.py" in s]
        par = ev.cpu_parent
        chain = []
        while par is not None and len(chain) < 3:
            chain.append(par.name[:48]); par = par.cpu_parent
        key = (ev.name + " " + str(ev.input_shapes)[:40], st[0].split("_amd/")[-1] if st else ("parents: " + " < ".join(chain)))
        acc[key] += 1
for (name, where), n in acc.most_common(45):
    print(f"{n:5d}  {name:18s} {where[:150]}")
