#!/bin/bash
# Functional rehearsal of bench.py's N > 1 path: 2 ranks on ONE GPU over gloo at C0 (not a scaling number).
# usage: gpurun -- 'bash scripts/gpu_dp2_rehearsal.sh <tag> [ENV=VAL ...]'
cd $GRAFT_REPO_ROOT
TAG=$1; shift
( for e in "$@"; do export $e; done
  EDRL_DIST_BACKEND=gloo EDRL_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --config C0 --steps 3 --warmup 1 > gpurun_out/dp2_$TAG.json 2> gpurun_out/dp2_$TAG.err )
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/dp2_$TAG.json') if l.startswith('{')][-1]); g=d['grad_exchange']
print('$TAG', '$*', d['value'], d['ms_per_step'], 'bwd_gpu', g['backward_gpu_ms'], g.get('launch_gpu_ms_after_backward_start'))"
