set -o pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q -s --timeout 300 -p no:cacheprovider -x > gpurun_out/t10.log 2>&1; rc=$?; echo "pytest exit=$rc" >> gpurun_out/t10.log
tail -2 gpurun_out/t10.log
if [ $rc -ne 0 ]; then exit 1; fi
timeout -k 10 600 python bench.py > gpurun_out/bench_c1_v4.json 2> gpurun_out/bench_c1_v4.err; echo "bench exit=$?" >> gpurun_out/bench_c1_v4.err
cat gpurun_out/bench_c1_v4.json; tail -2 gpurun_out/bench_c1_v4.err
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_c1_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/pmc_c1_$c.log 2>&1; echo "pmc $c exit=$?" >> $R/gpurun_out/pmc_c1_$c.log
tail -1 $R/gpurun_out/pmc_c1_$c.log
done
cd $R && python scripts/pmc_traffic.py gpurun_out/pmc_c1_FETCH_SIZE gpurun_out/pmc_c1_WRITE_SIZE gpurun_out/pmc_traffic_c1.json "C1: B=32/GPU, ResNet-50 encoders, 224x224 fundus + 32-slice OCT, fp32"
