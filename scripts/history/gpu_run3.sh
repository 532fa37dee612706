set -o pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider > gpurun_out/t3.log 2>&1; echo "pytest exit=$?" >> gpurun_out/t3.log
tail -3 gpurun_out/t3.log
timeout -k 10 900 python bench.py > gpurun_out/bench_c1.log 2>&1; echo "bench exit=$?" >> gpurun_out/bench_c1.log
tail -2 gpurun_out/bench_c1.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b8 -- python3 $R/bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_b8.log 2>&1; echo "prof exit=$?" >> $R/gpurun_out/prof_b8.log
tail -2 $R/gpurun_out/prof_b8.log
find $R/gpurun_out/prof_b8 -name "*stats*" | head
