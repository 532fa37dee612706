# K-split of small grids (EDRL_GATHER_TAIL_SPLIT=2) against tails only (=1): fp32 + head test files, then C1 and C0-sized steps
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layerwise.py tests/test_gpu_encoder.py tests/test_gpu_head.py tests/test_gpu_eval.py tests/test_gpu_vol3d.py tests/test_gpu_canary.py -x -q 2>&1 | tail -4 || exit 1
B="python bench.py --steps 6 --warmup 2 --no-cpu-baseline --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs"
for cfg in C1 C0; do
for sp in 1 2 1 2; do
  EDRL_GATHER_TAIL_SPLIT=$sp timeout -k 10 300 $B --config $cfg > gpurun_out/r4v_$cfg_sp$sp.json 2> gpurun_out/r4v_$cfg_sp$sp.err || { tail -5 gpurun_out/r4v_$cfg_sp$sp.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4v_$cfg_sp$sp.json').read().strip().splitlines()[-1])
print('$cfg split $sp', d['value'], d['ms_per_step'], d['roofline']['launches'], {k:round(v['ms_total']/d['steps'],2) for k,v in d['kernels'].items() if 'gather' in k})
PY
done
done
