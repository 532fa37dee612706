set -o pipefail
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof_c1.json 2> $R/gpurun_out/prof_c1.err; echo "prof exit=$?" >> $R/gpurun_out/prof_c1.err
cat $R/gpurun_out/prof_c1.json | cut -c1-200
