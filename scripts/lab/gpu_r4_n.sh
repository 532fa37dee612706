#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -x -s -k "stem" > gpurun_out/r4n_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/r4n_pytest.log; tail -3 gpurun_out/r4n_pytest.log; grep "stem wgrad" gpurun_out/r4n_pytest.log | tail -8
[ $rc -eq 0 ] || exit $rc
for v in 0 1; do
  EDRL_BF16_STEM_WGRAD_MMA=$v timeout -k 10 300 python bench.py --config C2 --steps 6 --warmup 2 --no-cpu-baseline --no-recompute-leg --in-order > gpurun_out/r4n_c2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r4n_c2.json"))
print("stem_wgrad_mma=$v value", d["value"], "ms", d["ms_per_step"], {k:round(x["ms_total"]/6,1) for k,x in d["kernels"].items() if "wgrad" in k})
PY
done
exit 0
