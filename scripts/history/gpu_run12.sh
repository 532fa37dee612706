set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/t12.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/t12.log
tail -2 gpurun_out/t12.log
if [ $rc -ne 0 ]; then exit 1; fi
for v in 0 1; do
EDRL_WGRAD_STREAM=$v timeout -k 10 600 python bench.py --no-cpu-baseline --no-kernel-timing > gpurun_out/bench_c1_ws$v.json 2> gpurun_out/bench_c1_ws$v.err; echo "bench exit=$?" >> gpurun_out/bench_c1_ws$v.err
python -c "
import json; d=json.load(open('gpurun_out/bench_c1_ws$v.json')); print('wgrad_stream=$v', d['value'], 'img/s', d['ms_per_step'], 'ms', d['peak_mem_GiB'], 'GiB')"
done
