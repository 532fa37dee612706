#!/bin/bash
# round-4 record runs: 50-step lines (SURVEY 8d protocol) for C1 and C2, C4 line, per-layer weight-gradient table
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs > gpurun_out/r4q_c1_50.json 2>/dev/null; echo "c1 rc=$?"
timeout -k 10 300 python bench.py --config C2 --steps 50 --warmup 5 --no-cpu-baseline --no-recompute-leg > gpurun_out/r4q_c2_50.json 2>/dev/null; echo "c2 rc=$?"
timeout -k 10 300 python bench.py --config C4 --steps 6 --warmup 2 --no-cpu-baseline --no-recompute-leg > gpurun_out/r4q_c4.json 2>/dev/null; echo "c4 rc=$?"
python - <<PY
import json
for f in ("r4q_c1_50","r4q_c2_50","r4q_c4"):
    d=json.load(open(f"gpurun_out/{f}.json"))
    print(f, d["value"], d["ms_per_step"], d["peak_mem_GiB"], d["roofline"]["frac"], d.get("view_overlap",{}).get("value"))
PY
timeout -k 10 300 python scripts/wgrad_layer_bench.py 2112 > gpurun_out/r4q_wgrad_layers.txt 2>&1; tail -3 gpurun_out/r4q_wgrad_layers.txt | cut -c1-200
exit 0
