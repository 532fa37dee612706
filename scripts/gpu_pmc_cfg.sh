# rocprofv3 --pmc passes over one bench step of a config (separate passes, kernel-trace only): HBM traffic, MFMA busy, instruction mix
# and LDS counters per kernel family.  usage: bash scripts/gpu_pmc_cfg.sh C2   -> gpurun_out/pmc_traffic_<cfg>.json
set -o pipefail
CFG=${1:-C2}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf $R/gpurun_out/pmc_${CFG}_$tag
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${CFG}_$tag -- python3 $R/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs --no-f32-mfma-leg > $R/gpurun_out/pmc_${CFG}_$tag.log 2>&1; rc=$?; echo "pmc $tag exit=$rc" >> $R/gpurun_out/pmc_${CFG}_$tag.log
  tail -1 $R/gpurun_out/pmc_${CFG}_$tag.log
  [ $rc -eq 0 ] || exit $rc
done
DESC=$(python3 -c "import sys; sys.path.insert(0,'$R'); import bench; print(bench.CONFIGS['$CFG'][5])")
cd $R && python scripts/pmc_traffic.py gpurun_out/pmc_${CFG}_FETCH_SIZE gpurun_out/pmc_${CFG}_WRITE_SIZE gpurun_out/pmc_traffic_${CFG}.json "$DESC" gpurun_out/pmc_${CFG}_SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/pmc_${CFG}_SQ_INSTS_VALU
# raw counter CSVs are tens of MB each; gpurun copies back at most 64 MiB: keep the reduced JSON and the logs only
rm -rf gpurun_out/pmc_${CFG}_FETCH_SIZE gpurun_out/pmc_${CFG}_WRITE_SIZE gpurun_out/pmc_${CFG}_SQ_VALU_MFMA_BUSY_CYCLES gpurun_out/pmc_${CFG}_SQ_INSTS_VALU
