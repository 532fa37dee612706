set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py tests/test_gpu_fullsize.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/t14.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/t14.log
tail -2 gpurun_out/t14.log
if [ $rc -ne 0 ]; then grep -E "^E  " gpurun_out/t14.log | head; exit 1; fi
for v in 3 1; do
EDRL_GATHER_VARIANT=$v timeout -k 10 300 python scripts/conv_layer_bench.py 1024 > gpurun_out/layers_1024_f$v.log 2>&1
echo "variant $v: $(tail -1 gpurun_out/layers_1024_f$v.log)"
done
grep -E "l1 3x3 64 |l2 3x3 128 |l3 1x1 256-1024|l4 1x1 2048-512|l1 1x1 64-256" gpurun_out/layers_1024_f1.log
