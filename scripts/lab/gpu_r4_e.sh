#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_canary.py -m gpu -q --timeout 600 -p no:cacheprovider -s -x > gpurun_out/r4e_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/r4e_pytest.log; tail -4 gpurun_out/r4e_pytest.log
grep -E "one-pass|bf16-MMA stem" gpurun_out/r4e_pytest.log | tail -16
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_head.py tests/test_gpu_fullsize.py -m gpu -q --timeout 600 -p no:cacheprovider -k "bf16 or c2 or c4 or C2 or C4" > gpurun_out/r4e_pytest2.log 2>&1
echo "pytest2 exit=$?" >> gpurun_out/r4e_pytest2.log; tail -3 gpurun_out/r4e_pytest2.log
for v in "EDRL_BF16_K64_BWD=0 EDRL_BF16_STEM_MMA=0" "EDRL_BF16_K64_BWD=0 EDRL_BF16_STEM_MMA=1" "EDRL_BF16_K64_BWD=1 EDRL_BF16_STEM_MMA=1"; do
  env $v timeout -k 10 300 python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline --no-recompute-leg --in-order > gpurun_out/r4e_c2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r4e_c2.json"))
print("$v value", d["value"], "ms", d["ms_per_step"], {k:round(x["ms_total"]/5,1) for k,x in d["kernels"].items()})
PY
done
exit 0
