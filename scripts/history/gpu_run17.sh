set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/t17.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/t17.log
tail -2 gpurun_out/t17.log
if [ $rc -ne 0 ]; then grep -E "^E  " gpurun_out/t17.log | head; exit 1; fi
timeout -k 10 300 python scripts/conv_layer_bench.py 1024 > gpurun_out/layers_1024_w2.log 2>&1
tail -1 gpurun_out/layers_1024_w2.log; grep -E "l1 3x3 64 |l2 3x3 128 |l3 3x3 256|l4 3x3 512 |l3 1x1 256-1024" gpurun_out/layers_1024_w2.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/bench_c1_v7.json 2> gpurun_out/bench_c1_v7.err; echo "bench exit=$?" >> gpurun_out/bench_c1_v7.err
python -c "
import json; d=json.load(open('gpurun_out/bench_c1_v7.json')); print(d['value'], 'img/s', d['ms_per_step'], 'ms', d['roofline']['achieved'], 'TF frac', d['roofline']['frac'], {k:(v['ms_total'],v['tflops']) for k,v in d['kernels'].items()})"
