#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -x -k "stem or pool or trunk" > gpurun_out/r4o_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/r4o_pytest.log; tail -3 gpurun_out/r4o_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --config C2 --steps 6 --warmup 2 --no-cpu-baseline --no-recompute-leg --in-order > gpurun_out/r4o_c2.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r4o_c2.json"))
print("value", d["value"], "ms", d["ms_per_step"], {k:(round(x["ms_total"]/6,1), x.get("GBps")) for k,x in d["kernels"].items() if "pool" in k})
PY
exit 0
