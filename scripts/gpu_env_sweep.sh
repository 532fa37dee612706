#!/bin/bash
# One-box sweep of environment settings over a bench command.  usage: gpurun -- 'bash scripts/gpu_env_sweep.sh <tag> "<bench args>" "<env A>" "<env B>" ...'
# ("-" = no extra environment); every setting runs once per round, two rounds, alternating.
set -o pipefail
cd $GRAFT_REPO_ROOT
TAG=$1; ARGS=$2; shift 2
for r in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    ( [ "$e" = "-" ] || export $e; timeout -k 10 300 python3 bench.py $ARGS > gpurun_out/${TAG}_${i}_$r.json 2> gpurun_out/${TAG}_${i}_$r.err ) || { echo "FAILED $e"; tail -3 gpurun_out/${TAG}_${i}_$r.err; exit 1; }
    python3 -c "import json; d=json.loads([l for l in open('gpurun_out/${TAG}_${i}_$r.json') if l.startswith('{')][-1]); print('round $r  [$e]', d['value'], d['ms_per_step'], d['peak_mem_GiB'])"
  done
done
