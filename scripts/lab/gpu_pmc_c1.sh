# rocprofv3 --pmc passes over the C1 bench step (separate passes, kernel-trace only): HBM traffic and MFMA busy per kernel family.
set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_c1_$tag
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c1_$tag -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs > $R/gpurun_out/pmc_c1_$tag.log 2>&1; rc=$?; echo "pmc $tag exit=$rc" >> $R/gpurun_out/pmc_c1_$tag.log
  tail -1 $R/gpurun_out/pmc_c1_$tag.log
  [ $rc -eq 0 ] || exit $rc
done
cd $R && python scripts/pmc_traffic.py gpurun_out/pmc_c1_FETCH_SIZE gpurun_out/pmc_c1_WRITE_SIZE gpurun_out/pmc_traffic_c1.json "C1: B=32/GPU, ResNet-50 encoders, 224x224 fundus + 32-slice OCT, fp32" gpurun_out/pmc_c1_SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/pmc_c1_SQ_INSTS_VALU
find gpurun_out/pmc_c1_* -name "*.csv" -size +20M -delete
