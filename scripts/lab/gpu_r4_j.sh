#!/bin/bash
# round-4 consolidation: full suite, default bench line, kernel-stats profiles of C1 / C2, PMC passes of C2 / C1
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/r4j_pytest.log 2>&1
rc=$?; echo "pytest exit=$rc" >> gpurun_out/r4j_pytest.log; tail -3 gpurun_out/r4j_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r4j_smoke.log 2>&1; tail -1 gpurun_out/r4j_smoke.log
SECONDS=0
timeout -k 10 500 python bench.py > gpurun_out/r4j_bench.json 2> gpurun_out/r4j_bench.err
echo "bench exit=$? wall=${SECONDS}s"; cut -c1-300 gpurun_out/r4j_bench.json
bash scripts/gpu_prof.sh C2 > gpurun_out/r4j_prof_c2.log 2>&1; tail -2 gpurun_out/r4j_prof_c2.log
bash scripts/gpu_prof.sh C1 > gpurun_out/r4j_prof_c1.log 2>&1; tail -2 gpurun_out/r4j_prof_c1.log
bash scripts/gpu_pmc_cfg.sh C2 > gpurun_out/r4j_pmc_c2.log 2>&1; tail -3 gpurun_out/r4j_pmc_c2.log
bash scripts/gpu_pmc_cfg.sh C1 > gpurun_out/r4j_pmc_c1.log 2>&1; tail -3 gpurun_out/r4j_pmc_c1.log
exit 0
