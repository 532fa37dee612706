"""One traced training step of a bench.py configuration, for call-by-call attribution of a rocprofv3 pass.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 scripts/step_trace.py --config C2 --out calls.json

Builds the model / optimiser / resident synthetic batch exactly as bench.py does, runs `--warmup` untraced steps, then ONE step
with the library's call tracer on (edrl_amd._lib.trace_begin): every library call is preceded by an empty marker dispatch
(`edrl_trace_mark_kernel`) and logged with its scalar arguments and the algorithmic flops / bytes its wrapper declares.  The
k-th marker row of the trace then precedes the kernels of the k-th record of `--out` (scripts/trace_traffic.py does the join).
The step runs in order on one stream (no view overlap), so dispatch order = call order."""
import argparse
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="C2")
ap.add_argument("--batch", type=int, default=0)
ap.add_argument("--warmup", type=int, default=1)
ap.add_argument("--out", required=True)
a = ap.parse_args()

import torch
import bench
import edrl_amd

B, depth, HW, S, enc_dtype, desc = bench.CONFIGS[a.config]
if a.batch:
    B = a.batch
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
args = types.SimpleNamespace(mode="train", batch_size=B, encoder_depth=depth, activation_recompute=a.config in bench.RECOMPUTE,
                             encoder_dtype=enc_dtype, oct_encoder="3d" if a.config.endswith("-3D") else "slices", oct3d_depth=18)
torch.manual_seed(0)
model = edrl_amd.MedFusion(2, 2, None, args).to(dev).train()
opt = edrl_amd.FusedAdam(model.parameters(), lr=1e-4, weight_decay=1e-6)
data, y = edrl_amd.synthetic_batch(B, HW, HW, S, device=dev, seed=1234, rank=0, drop_oct_high=(a.config == "C4"))
edrl_amd.set_view_overlap(False)
for _ in range(a.warmup):
    edrl_amd.train_step(model, opt, data, y)
torch.cuda.synchronize()
edrl_amd._lib.trace_begin()
out = edrl_amd.train_step(model, opt, data, y)
torch.cuda.synchronize()
calls = edrl_amd._lib.trace_end()
loss = out["loss"].item()
assert loss == loss, "NaN loss"
json.dump({"config": a.config, "workload": desc, "per_gpu_batch": B, "calls": calls}, open(a.out, "w"))
print(f"step_trace: {len(calls)} library calls traced, loss {loss:.6f}")
