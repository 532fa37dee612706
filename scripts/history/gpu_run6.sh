set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider -x > gpurun_out/t6.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/t6.log
tail -3 gpurun_out/t6.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 300 python scripts/conv_layer_bench.py 256 > gpurun_out/layers_256_v3.log 2>&1; echo "layers exit=$?" >> gpurun_out/layers_256_v3.log
cat gpurun_out/layers_256_v3.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/bench_c1_v3.json 2> gpurun_out/bench_c1_v3.err; echo "bench exit=$?" >> gpurun_out/bench_c1_v3.err
cat gpurun_out/bench_c1_v3.json; tail -2 gpurun_out/bench_c1_v3.err
