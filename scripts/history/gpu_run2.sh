set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider > gpurun_out/t2.log 2>&1; echo "pytest exit=$?" >> gpurun_out/t2.log
timeout -k 10 120 python __graft_entry__.py smoke > gpurun_out/smoke2.log 2>&1; echo "smoke exit=$?" >> gpurun_out/smoke2.log
timeout -k 10 400 python bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench2_b8.log 2>&1; echo "bench exit=$?" >> gpurun_out/bench2_b8.log
tail -3 gpurun_out/t2.log; tail -2 gpurun_out/smoke2.log; tail -3 gpurun_out/bench2_b8.log
