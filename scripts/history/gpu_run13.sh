set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s --timeout 300 -p no:cacheprovider > gpurun_out/t13.log 2>&1; echo "pytest exit=$?" >> gpurun_out/t13.log
tail -3 gpurun_out/t13.log; grep -E "^E  |parity\]" gpurun_out/t13.log | head
