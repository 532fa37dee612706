# K-split tail of the fp32 gather: fp32 test files, then C1 with the switch off / on
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layerwise.py tests/test_gpu_encoder.py tests/test_gpu_head.py tests/test_gpu_vol3d.py tests/test_gpu_canary.py -x -q 2>&1 | tail -6 || exit 1
B="python bench.py --config C1 --steps 6 --warmup 2 --no-cpu-baseline --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs"
for sp in 0 1 0 1; do
  EDRL_GATHER_TAIL_SPLIT=$sp timeout -k 10 300 $B > gpurun_out/r4u_sp$sp.json 2> gpurun_out/r4u_sp$sp.err || { tail -5 gpurun_out/r4u_sp$sp.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4u_sp$sp.json').read().strip().splitlines()[-1])
print('split $sp', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['launches'], {k:round(v['ms_total']/d['steps'],1) for k,v in d['kernels'].items()})
PY
done
