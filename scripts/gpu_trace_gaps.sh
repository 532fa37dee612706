#!/bin/bash
# usage: gpurun -- 'bash scripts/gpu_trace_gaps.sh C1'  -> gpurun_out/gaps_<cfg>.txt (scripts/trace_gaps.py over an in-order bench step)
set -o pipefail
CFG=${1:-C1}; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/gaps_$CFG
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gaps_$CFG -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs --no-f32-mfma-leg > $R/gpurun_out/gaps_$CFG.log 2>&1 || { tail -3 $R/gpurun_out/gaps_$CFG.log; exit 1; }
cd $R && python3 scripts/trace_gaps.py gpurun_out/gaps_$CFG > gpurun_out/gaps_$CFG.txt && cat gpurun_out/gaps_$CFG.txt
rm -rf gpurun_out/gaps_$CFG
