# 3-D path: volume tests + the C1-3D line
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests/test_gpu_vol3d.py -x -q 2>&1 | tail -8 || exit 1
timeout -k 10 400 python bench.py --config C1-3D --steps 5 --warmup 2 --no-cpu-baseline --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs > gpurun_out/r4t_c1_3d.json 2> gpurun_out/r4t_c1_3d.err || { tail -5 gpurun_out/r4t_c1_3d.err; exit 1; }
python - <<PY
import json
d=json.loads(open('gpurun_out/r4t_c1_3d.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], {k:round(v['ms_total']/d['steps'],1) for k,v in d['kernels'].items()})
PY
