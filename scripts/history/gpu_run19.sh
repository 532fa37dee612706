set -o pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/t19.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/t19.log
tail -2 gpurun_out/t19.log
if [ $rc -ne 0 ]; then grep -E "^E  " gpurun_out/t19.log | head; exit 1; fi
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/bench_c1_v8.json 2> gpurun_out/bench_c1_v8.err; echo "bench exit=$?" >> gpurun_out/bench_c1_v8.err
python -c "
import json; d=json.load(open('gpurun_out/bench_c1_v8.json')); print(d['value'], 'img/s', d['ms_per_step'], 'ms', d['roofline']['achieved'], 'TF frac', d['roofline']['frac'], {k:(v['ms_total'],v['tflops']) for k,v in d['kernels'].items()})"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c1b -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_c1b.json 2> $R/gpurun_out/prof_c1b.err; echo "prof exit=$?" >> $R/gpurun_out/prof_c1b.err
