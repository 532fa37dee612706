# A/B: stage-2 bottleneck blocks (128 planes) on the "wide" scheme instead of the fused one (EDRL_BF16_FUSE_MAXPLANES 128 -> 64)
set -o pipefail
cd $GRAFT_REPO_ROOT
B="python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs"
for mp in 128 64 128 64; do
  EDRL_BF16_FUSE_MAXPLANES=$mp timeout -k 10 300 $B > gpurun_out/r4r_mp$mp.json 2> gpurun_out/r4r_mp$mp.err || { tail -5 gpurun_out/r4r_mp$mp.err; exit 1; }
  python - <<PY
import json
d=json.loads(open('gpurun_out/r4r_mp$mp.json').read().strip().splitlines()[-1])
print('maxplanes $mp', d['value'], d['ms_per_step'], {k:round(v['ms_total']/5,1) for k,v in d['kernels'].items()})
PY
done
