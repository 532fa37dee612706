#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "v3" > gpurun_out/r4i_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/r4i_pytest.log; tail -3 gpurun_out/r4i_pytest.log
[ $rc -eq 0 ] || exit $rc
for v in "EDRL_BF16_V3_PERSIST=1 EDRL_V3_FWD_KMIN=512" "EDRL_BF16_V3_PERSIST=1 EDRL_V3_FWD_KMIN=256" "EDRL_BF16_V3_PERSIST=2 EDRL_V3_FWD_KMIN=512" "EDRL_BF16_V3_PERSIST=2 EDRL_V3_FWD_KMIN=256"; do
  env $v timeout -k 10 300 python bench.py --config C2 --steps 6 --warmup 2 --no-cpu-baseline --no-recompute-leg --in-order > gpurun_out/r4i_c2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r4i_c2.json"))
print("$v value", d["value"], "ms", d["ms_per_step"], {k:round(x["ms_total"]/6,1) for k,x in d["kernels"].items() if "bf16" in k})
PY
done
exit 0
