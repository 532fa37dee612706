set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/t15.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/t15.log
tail -2 gpurun_out/t15.log
if [ $rc -ne 0 ]; then grep -E "^E  " gpurun_out/t15.log | head; exit 1; fi
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/bench_c1_v5.json 2> gpurun_out/bench_c1_v5.err; echo "bench exit=$?" >> gpurun_out/bench_c1_v5.err
python -c "
import json; d=json.load(open('gpurun_out/bench_c1_v5.json')); print(d['value'], 'img/s', d['ms_per_step'], 'ms', d['peak_mem_GiB'], 'GiB', d['roofline']['achieved'], 'TF', {k:(v['ms_total'],v['tflops']) for k,v in d['kernels'].items()})"
