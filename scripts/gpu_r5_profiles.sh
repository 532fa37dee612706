#!/bin/bash
# Round-5 profile set of one config: rocprofv3 --kernel-trace --stats of the in-order bench step + the four PMC passes.
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r5_profiles.sh C4'   -> gpurun_out/prof_<cfg>_kernel_stats.csv, pmc_traffic_<cfg>.json
set -o pipefail
CFG=${1:-C2}
cd $GRAFT_REPO_ROOT
bash scripts/gpu_prof.sh $CFG > gpurun_out/r5_prof_$CFG.log 2>&1 || { tail -5 gpurun_out/r5_prof_$CFG.log; exit 1; }
tail -3 gpurun_out/r5_prof_$CFG.log
bash scripts/gpu_pmc_cfg.sh $CFG > gpurun_out/r5_pmc_$CFG.log 2>&1 || { tail -5 gpurun_out/r5_pmc_$CFG.log; exit 1; }
tail -12 gpurun_out/r5_pmc_$CFG.log
