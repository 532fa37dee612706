#!/bin/bash
# fp32 trunk block policy (encoders._K32: EDRL_F32_MID_SEP / EDRL_F32_FUSE_MAXPLANES) on one box: C1 in-order rate per setting.
# usage: gpurun -- 'bash scripts/gpu_f32_policy_sweep.sh'
set -o pipefail
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift; ( export "$@"; timeout -k 10 300 python3 bench.py --config C1 --in-order --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-recompute-leg --no-anchor-leg --no-bf16-legs --no-f32-mfma-leg > gpurun_out/pol_$tag.json 2> gpurun_out/pol_$tag.err ) || { echo "FAILED $tag"; tail -5 gpurun_out/pol_$tag.err; return 1; }
  python3 -c "import json; d=json.load(open('gpurun_out/pol_$tag.json')); print('$tag', '$*', d['value'], d['ms_per_step'], d['peak_mem_GiB'])"; }
run default X=1 || exit 1
run mp128 EDRL_F32_FUSE_MAXPLANES=128 || exit 1
run mp64 EDRL_F32_FUSE_MAXPLANES=64 || exit 1
run nowide EDRL_F32_FUSE_MAXPLANES=1073741824 || exit 1
run allfused EDRL_F32_FUSE_MAXPLANES=1073741824 EDRL_F32_MID_SEP=0 || exit 1
run default2 X=1 || exit 1
