#!/bin/bash
# Full GPU gate: pytest -m gpu, smoke(), default bench line.  Usage: gpurun --timeout 1200 -- 'bash scripts/gpu_full_suite.sh'
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 700 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/suite_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/suite_pytest.log; tail -4 gpurun_out/suite_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/suite_smoke.log 2>&1
rc=$?; tail -2 gpurun_out/suite_smoke.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python bench.py > gpurun_out/suite_bench.json 2> gpurun_out/suite_bench.err
rc=$?; tail -2 gpurun_out/suite_bench.err; cat gpurun_out/suite_bench.json
exit $rc
