#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_head.py tests/test_gpu_kernels.py tests/test_gpu_layerwise.py tests/test_gpu_canary.py tests/test_gpu_encoder.py tests/test_gpu_bf16.py -m gpu -q --timeout 600 -p no:cacheprovider -s > gpurun_out/r4d_pytest.log 2>&1
echo "pytest exit=$?" >> gpurun_out/r4d_pytest.log; tail -4 gpurun_out/r4d_pytest.log
grep -E "worst trunk|Frobenius error|bf16-MMA stem" gpurun_out/r4d_pytest.log | tail -16
timeout -k 10 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-recompute-leg --no-anchor-leg --no-bf16-legs > gpurun_out/r4d_c1.json 2>/dev/null
python - <<PY
import json
d=json.load(open("gpurun_out/r4d_c1.json"))
print("C1 value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "overlap", d.get("view_overlap",{}).get("value"), {k:round(x["ms_total"]/8,1) for k,x in d["kernels"].items()})
PY
for v in 0 1; do
  EDRL_BF16_STEM_MMA=$v timeout -k 10 300 python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline --no-recompute-leg --in-order > gpurun_out/r4d_c2_stem$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r4d_c2_stem$v.json"))
print("stem_mma=$v value", d["value"], "ms", d["ms_per_step"], {k:round(x["ms_total"]/5,1) for k,x in d["kernels"].items()})
PY
done
exit 0
