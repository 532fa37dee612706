#!/bin/bash
# Round-5 evidence set on the final kernels: per-layer traffic tables (C2, C1) and the rocprofv3 stats + PMC passes of C1, C2, C4.
# usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r5_evidence.sh'
set -o pipefail
cd $GRAFT_REPO_ROOT
for c in C2 C1; do bash scripts/gpu_trace_layers.sh $c > gpurun_out/r5_trace_$c.log 2>&1 || { tail -5 gpurun_out/r5_trace_$c.log; exit 1; }; done
for c in C1 C2 C4; do bash scripts/gpu_r5_profiles.sh $c > gpurun_out/r5_profiles_$c.log 2>&1 || { tail -5 gpurun_out/r5_profiles_$c.log; exit 1; }; echo "profiles $c done"; done
