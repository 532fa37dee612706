#!/bin/bash
# final check of the round: full GPU suite, smoke, default bench line
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r4f_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/r4f_pytest.log; tail -3 gpurun_out/r4f_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4f_smoke.log 2>&1; tail -1 gpurun_out/r4f_smoke.log
SECONDS=0
timeout -k 10 500 python bench.py > gpurun_out/r4f_bench.json 2> gpurun_out/r4f_bench.err
echo "bench exit=$? wall=${SECONDS}s"; cut -c1-260 gpurun_out/r4f_bench.json
