#!/bin/bash
# Round-5 gate: full GPU suite, smoke, the default bench line.  usage: gpurun --timeout 1200 -- 'bash scripts/gpu_r5_suite.sh <tag>'
set -o pipefail
TAG=${1:-r5}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/${TAG}_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/${TAG}_pytest.log; tail -5 gpurun_out/${TAG}_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -1 gpurun_out/${TAG}_smoke.log
SECONDS=0
timeout -k 10 500 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
echo "bench exit=$? wall=${SECONDS}s"; cut -c1-300 gpurun_out/${TAG}_bench.json
