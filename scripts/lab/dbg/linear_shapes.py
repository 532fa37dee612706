"""Which Linear launches dominate the head at the C1 shapes: per (kind, rows, K, N) time from HIP events (diagnostic)."""
import sys, os, types, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, edrl_amd
ops = edrl_amd.ops
dev = torch.device("cuda:0")
B = 32
args = types.SimpleNamespace(mode="train", batch_size=B, encoder_depth=50)
torch.manual_seed(0)
m = edrl_amd.MedFusion(2, 2, None, args).to(dev).train()
# head only: tokens of the C1 shapes (224x224 -> 7x7 = 49 fundus tokens, 32 OCT tokens)
rec = collections.defaultdict(list)
orig = ops._launch_timed
def spy(kind, flops, name, *a, **kw):
    if not kind.startswith("linear"):
        return orig(kind, flops, name, *a, **kw)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); ops.L.call(name, *a); e1.record()
    if name.endswith("fwd_f32"):   key = ("fwd", a[5], a[8], a[11])        # rows, cin, cout
    elif name.endswith("dgrad_f32"): key = ("dgrad", a[3], a[9], a[6])     # rows, K=cout, N=cin
    else: key = ("wgrad", a[5], a[11], a[8])
    rec[key].append((e0, e1))
ops._launch_timed = spy
data, y = edrl_amd.synthetic_batch(B, 224, 224, 32, device=dev)
opt = edrl_amd.FusedAdam(m.parameters(), lr=1e-4)
for it in range(2):
    rec.clear()
    edrl_amd.train_step(m, opt, data, y)
torch.cuda.synchronize()
rows = sorted(((sum(a.elapsed_time(b) for a, b in v), k, len(v)) for k, v in rec.items()), reverse=True)
tot = sum(r[0] for r in rows)
print(f"linear total {tot:.2f} ms per step, {sum(r[2] for r in rows)} launches")
for ms, k, n in rows[:25]:
    kind, r_, K, N = k
    print(f"{ms:7.3f} ms  x{n:3d}  {kind:6s} rows {r_:6d} K {K:5d} N {N:5d}   {2.0*r_*K*N*n/ms/1e9:7.1f} TFLOP/s")
