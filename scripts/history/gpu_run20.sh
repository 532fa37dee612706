set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/t20.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/t20.log
tail -2 gpurun_out/t20.log
if [ $rc -ne 0 ]; then grep -E "^E  " gpurun_out/t20.log | head; exit 1; fi
EDRL_FUNDUS_STREAM=1 timeout -k 10 300 python -m pytest tests/test_gpu_head.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k full_train > gpurun_out/t20b.log 2>&1; echo "fundus-stream test exit=$?" >> gpurun_out/t20b.log; tail -2 gpurun_out/t20b.log
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke20.log 2>&1; echo "smoke exit=$?" >> gpurun_out/smoke20.log; tail -2 gpurun_out/smoke20.log
for v in 0 1; do
EDRL_FUNDUS_STREAM=$v timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/bench_c1_fs$v.json 2> gpurun_out/bench_c1_fs$v.err; echo "bench exit=$?" >> gpurun_out/bench_c1_fs$v.err
python -c "
import json; d=json.load(open('gpurun_out/bench_c1_fs$v.json')); print('fundus_stream=$v', d['value'], 'img/s', d['ms_per_step'], 'ms', d['roofline']['achieved'], 'TF frac', d['roofline']['frac'])"
done
