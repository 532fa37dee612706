#!/bin/bash
# round-4 first GPU pass: full suite (with the new large-mean / diagnostics tests), default bench line (new legs), C2 stream experiments
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -s > gpurun_out/r4a_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/r4a_pytest.log; tail -5 gpurun_out/r4a_pytest.log
grep -E "large-mean|relative Frobenius|DP1 exchange|full step" gpurun_out/r4a_pytest.log | tail -40
timeout -k 10 500 python bench.py > gpurun_out/r4a_bench.json 2> gpurun_out/r4a_bench.err
echo "bench exit=$?"; tail -3 gpurun_out/r4a_bench.err; cut -c1-600 gpurun_out/r4a_bench.json
for v in "EDRL_WGRAD_STREAM=1" "EDRL_WGRAD_STREAM=1 EDRL_FUNDUS_STREAM=1"; do
  echo "== C2 $v"
  env $v timeout -k 10 300 python bench.py --config C2 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-recompute-leg > gpurun_out/r4a_c2_$(echo $v | tr ' =' '__').json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r4a_c2_$(echo $v | tr ' =' '__').json"))
print("value", d["value"], "ms", d["ms_per_step"], "overlap", d.get("view_overlap",{}).get("value"))
PY
done
exit 0
