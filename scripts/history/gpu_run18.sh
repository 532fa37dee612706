set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_encoder.py tests/test_gpu_fullsize.py -m gpu -q --timeout 300 -p no:cacheprovider -x > gpurun_out/t18.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/t18.log
tail -2 gpurun_out/t18.log
if [ $rc -ne 0 ]; then grep -E "^E  " gpurun_out/t18.log | head; exit 1; fi
timeout -k 10 300 python scripts/conv_layer_bench.py 1024 > gpurun_out/layers_1024_p.log 2>&1
tail -1 gpurun_out/layers_1024_p.log; grep -E "l1 3x3 64 |l2 3x3 128 |l3 3x3 256|l4 3x3 512 |l3 1x1 256-1024|l1 1x1 64-256|l2 1x1 256-128" gpurun_out/layers_1024_p.log
