set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python bench.py > gpurun_out/bench_c1_v6.json 2> gpurun_out/bench_c1_v6.err; echo "bench exit=$?" >> gpurun_out/bench_c1_v6.err
python -c "
import json; d=json.load(open('gpurun_out/bench_c1_v6.json')); print(d['value'], 'img/s', d['ms_per_step'], 'ms', d['peak_mem_GiB'], 'GiB', d['roofline']['achieved'], 'TF frac', d['roofline']['frac'], {k:(v['ms_total'],v['tflops']) for k,v in d['kernels'].items()}, d['cpu_baseline']['value'])"
