#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for v in 0 512 1100 0 512; do
  EDRL_V3_EPI_KMIN=$v timeout -k 10 300 python bench.py --config C2 --steps 6 --warmup 2 --no-cpu-baseline --no-recompute-leg --in-order > gpurun_out/r4p_c2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r4p_c2.json"))
print("epi_kmin=$v value", d["value"], "ms", d["ms_per_step"], {k:round(x["ms_total"]/6,1) for k,x in d["kernels"].items() if "gather_bf16" in k})
PY
done
exit 0
