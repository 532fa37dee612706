set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider -x > gpurun_out/t5.log 2>&1; echo "pytest exit=$?" >> gpurun_out/t5.log
tail -3 gpurun_out/t5.log
timeout -k 10 300 python scripts/conv_layer_bench.py 256 > gpurun_out/layers_256_v2.log 2>&1; echo "layers exit=$?" >> gpurun_out/layers_256_v2.log
cat gpurun_out/layers_256_v2.log
timeout -k 10 600 python bench.py --no-cpu-baseline > gpurun_out/bench_c1_v2.json 2> gpurun_out/bench_c1_v2.err; echo "bench exit=$?" >> gpurun_out/bench_c1_v2.err
cat gpurun_out/bench_c1_v2.json
