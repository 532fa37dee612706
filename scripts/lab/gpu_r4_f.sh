#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -s -x -k "one_pass or stem_bf16" > gpurun_out/r4f_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/r4f_pytest.log; tail -3 gpurun_out/r4f_pytest.log
[ $rc -eq 0 ] || exit $rc
for v in "EDRL_BF16_K64_BWD=1"; do
  env $v timeout -k 10 300 python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline --no-recompute-leg --in-order > gpurun_out/r4f_c2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r4f_c2.json"))
print("$v value", d["value"], "ms", d["ms_per_step"], {k:round(x["ms_total"]/5,1) for k,x in d["kernels"].items()})
PY
done
exit 0
