set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1
grep -ciE "MFMA" $R/gpurun_out/counters.txt
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS"; do
  tag=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python3 $R/scripts/one_layer.py 256 14 256 3 1 1 256 fwd > $R/gpurun_out/pmc_$tag.log 2>&1; echo "exit=$?" >> $R/gpurun_out/pmc_$tag.log
  tail -1 $R/gpurun_out/pmc_$tag.log
done
find $R/gpurun_out -name "*counter_collection.csv" | head
