#!/bin/bash
# Same-box A/B of the shipped library against an A/B build (csrc: `make ab AB_SRC=... AB_DEFS=...` -> libedrl_hip_ab.so), runs
# alternating.  usage: gpurun -- 'bash scripts/gpu_ab.sh <tag> <rounds> <command ...>'  -> gpurun_out/<tag>_{ship,ab}_<i>.txt
set -o pipefail
TAG=$1; ROUNDS=$2; shift 2
R=$GRAFT_REPO_ROOT
PKG=$(ls -d $R/*_amd)
mkdir -p $R/gpurun_out
for i in $(seq 1 $ROUNDS); do
  for arm in ship ab; do
    if [ $arm = ab ]; then export EDRL_LIB_PATH=$PKG/libedrl_hip_ab.so; else unset EDRL_LIB_PATH; fi
    timeout -k 10 400 "$@" > $R/gpurun_out/${TAG}_${arm}_$i.txt 2> $R/gpurun_out/${TAG}_${arm}_$i.err || { echo "FAILED $arm $i"; tail -5 $R/gpurun_out/${TAG}_${arm}_$i.err; exit 1; }
    echo "== $arm $i"; tail -3 $R/gpurun_out/${TAG}_${arm}_$i.txt | cut -c1-250
  done
done
