#!/bin/bash
# Stall / issue breakdown of single conv layers: rocprofv3 --pmc passes (kernel-trace only) over scripts/one_layer.py.
# usage: gpurun -- 'bash scripts/pmc_layers.sh "<Ci H Co k s p N mode>" ...'    -> gpurun_out/pmcL/<tag>.txt (scripts/pmc_layers_parse.py)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/pmcL; rm -rf $OUT; mkdir -p $OUT
PASSES=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
        "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" \
        "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" \
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT" \
        "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE" \
        "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES")
for spec in "$@"; do
  tag=$(echo $spec | tr ' ' '_')
  pi=0
  for pass in "${PASSES[@]}"; do
    REPS=4 timeout -k 10 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $OUT/$tag/p$pi -- python3 $R/scripts/one_layer.py $spec > $OUT/$tag.p$pi.log 2>&1 || { echo "FAILED $tag pass $pi"; tail -3 $OUT/$tag.p$pi.log; exit 1; }
    pi=$((pi+1))
  done
  python3 $R/scripts/pmc_layers_parse.py $OUT/$tag > $OUT/$tag.txt && cat $OUT/$tag.txt
  rm -rf $OUT/$tag
done
