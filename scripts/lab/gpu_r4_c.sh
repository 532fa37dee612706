#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
EDRL_TEST_GRAD_TABLE=1 timeout -k 10 600 python -m pytest tests/test_gpu_head.py -m gpu -q --timeout 600 -p no:cacheprovider -s -k "test_full_train_step_vs_oracle" > gpurun_out/r4c_step.log 2>&1
echo "pytest exit=$?" >> gpurun_out/r4c_step.log; tail -3 gpurun_out/r4c_step.log
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_head.py -m gpu -q --timeout 600 -p no:cacheprovider -k "not test_full_train_step_vs_oracle" > gpurun_out/r4c_rest.log 2>&1
echo "pytest exit=$?" >> gpurun_out/r4c_rest.log; tail -3 gpurun_out/r4c_rest.log
for v in 0 1; do
  EDRL_BF16_WIDE_SEP=$v timeout -k 10 300 python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline --no-recompute-leg --in-order > gpurun_out/r4b_c2_wide$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r4b_c2_wide$v.json"))
print("wide_sep=$v value", d["value"], "ms", d["ms_per_step"], "mem", d["peak_mem_GiB"], {k:round(x["ms_total"]/5,1) for k,x in d["kernels"].items()})
PY
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for CFG in C2 C1; do
  rm -rf $R/gpurun_out/trace_$CFG
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$CFG -- python3 $R/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs > /dev/null 2>&1
  python3 $R/scripts/trace_by_grid.py $R/gpurun_out/trace_$CFG 2 90 > $R/gpurun_out/r4b_${CFG}_by_grid.txt
  find $R/gpurun_out/trace_$CFG -name "*.csv" -delete
  head -3 $R/gpurun_out/r4b_${CFG}_by_grid.txt
done
exit 0
