#!/bin/bash
# Same-box alternating A/B of two environments.  usage: gpurun -- 'bash scripts/gpu_env_ab.sh <tag> <rounds> "<env B assignments>" <command ...>'
# arm A = the command as is, arm B = the command with the assignments exported (e.g. "EDRL_TRUNK_GRAD_STASH=0 EDRL_WEIGHT_SHADOWS=0").
set -o pipefail
TAG=$1; ROUNDS=$2; ENVB=$3; shift 3
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for i in $(seq 1 $ROUNDS); do
  for arm in a b; do
    if [ $arm = b ]; then
      ( export $ENVB; timeout -k 10 400 "$@" > $R/gpurun_out/${TAG}_${arm}_$i.txt 2> $R/gpurun_out/${TAG}_${arm}_$i.err ) || { echo "FAILED $arm $i"; tail -5 $R/gpurun_out/${TAG}_${arm}_$i.err; exit 1; }
    else
      timeout -k 10 400 "$@" > $R/gpurun_out/${TAG}_${arm}_$i.txt 2> $R/gpurun_out/${TAG}_${arm}_$i.err || { echo "FAILED $arm $i"; tail -5 $R/gpurun_out/${TAG}_${arm}_$i.err; exit 1; }
    fi
    echo "== $arm $i: $(python3 -c "import json,sys; d=json.load(open('$R/gpurun_out/${TAG}_${arm}_$i.txt')); print(d['value'], d['ms_per_step'])" 2>/dev/null)"
  done
done
