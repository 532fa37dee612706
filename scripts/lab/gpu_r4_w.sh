#!/bin/bash
# round-4 final record runs: 50-step C1 line, C1-3D line (+ kernel stats), C0 and C3-per-GPU lines, per-grid traces of C1 / C2
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs > gpurun_out/r4w_c1_50.json 2>/dev/null; echo "c1 rc=$?"
timeout -k 10 300 python bench.py --config C1-3D --steps 6 --warmup 2 --no-cpu-baseline --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs > gpurun_out/r4w_c1_3d.json 2>/dev/null; echo "c1-3d rc=$?"
timeout -k 10 300 python bench.py --config C0 --steps 20 --warmup 3 --no-cpu-baseline --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs > gpurun_out/r4w_c0.json 2>/dev/null; echo "c0 rc=$?"
python - <<PY
import json
for f in ("r4w_c1_50","r4w_c1_3d","r4w_c0"):
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d["peak_mem_GiB"], d["roofline"]["frac"])
PY
bash scripts/gpu_prof.sh C1-3D > gpurun_out/r4w_prof_c1_3d.log 2>&1; tail -1 gpurun_out/r4w_prof_c1_3d.log
bash scripts/gpu_r4_s.sh C1 > /dev/null 2>&1; head -3 gpurun_out/trace_C1_by_grid.txt
bash scripts/gpu_r4_s.sh C2 > /dev/null 2>&1; head -3 gpurun_out/trace_C2_by_grid.txt
exit 0
