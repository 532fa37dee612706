#!/bin/bash
# Shipped library (fp32 contractions as bf16x3 splits) against the fp32-MFMA build: accuracy per layer, then the layer bench.
# usage: gpurun -- 'bash scripts/gpu_split_probe.sh [images]'
set -o pipefail
R=$GRAFT_REPO_ROOT; PKG=$(ls -d $R/*_amd); cd $R
N=${1:-1056}
for arm in ship f32mfma; do
  if [ $arm = f32mfma ]; then export EDRL_LIB_PATH=$PKG/libedrl_hip_f32mfma.so; else unset EDRL_LIB_PATH; fi
  timeout -k 10 300 python3 scripts/split_accuracy.py 8 > gpurun_out/split_acc_$arm.txt 2>&1 || { tail -5 gpurun_out/split_acc_$arm.txt; exit 1; }
  cat gpurun_out/split_acc_$arm.txt
done
for arm in ship f32mfma; do
  if [ $arm = f32mfma ]; then export EDRL_LIB_PATH=$PKG/libedrl_hip_f32mfma.so; else unset EDRL_LIB_PATH; fi
  timeout -k 10 400 python3 scripts/conv_layer_bench.py $N > gpurun_out/split_layers_$arm.txt 2>&1 || { tail -5 gpurun_out/split_layers_$arm.txt; exit 1; }
  echo "== $arm"; cat gpurun_out/split_layers_$arm.txt
done
