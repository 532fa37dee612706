set -o pipefail
cd $GRAFT_REPO_ROOT
EDRL_GATHER_VARIANT=1 timeout -k 10 300 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "conv or linear" > gpurun_out/t8.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/t8.log
tail -3 gpurun_out/t8.log
if [ $rc -ne 0 ]; then exit 1; fi
EDRL_GATHER_VARIANT=1 timeout -k 10 300 python scripts/conv_layer_bench.py 1024 > gpurun_out/layers_1024_var1.log 2>&1; echo "layers exit=$?" >> gpurun_out/layers_1024_var1.log
cat gpurun_out/layers_1024_var1.log
