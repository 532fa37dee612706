set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
for pass in "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  EDRL_GATHER_VARIANT=$v timeout -k 10 200 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc2_v$v -- python3 $R/scripts/one_layer.py 128 28 128 3 1 1 1024 fwd > $R/gpurun_out/pmc2_v$v.log 2>&1; echo "exit=$?" >> $R/gpurun_out/pmc2_v$v.log
  tail -1 $R/gpurun_out/pmc2_v$v.log
done
done
