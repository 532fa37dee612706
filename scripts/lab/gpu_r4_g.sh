#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_canary.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "v3 or launchers" > gpurun_out/r4g_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/r4g_pytest.log; tail -4 gpurun_out/r4g_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python scripts/v3_layer_bench.py 2112 > gpurun_out/r4g_v3_layers.txt 2>&1
tail -20 gpurun_out/r4g_v3_layers.txt | cut -c1-210
for v in 0 1; do
  EDRL_BF16_V3_PERSIST=$v timeout -k 10 300 python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline --no-recompute-leg --in-order > gpurun_out/r4g_c2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r4g_c2.json"))
print("persist=$v value", d["value"], "ms", d["ms_per_step"], {k:round(x["ms_total"]/5,1) for k,x in d["kernels"].items()})
PY
done
exit 0
