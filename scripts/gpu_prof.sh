# usage: bash scripts/gpu_prof.sh C1|C2   -> gpurun_out/prof_<cfg>/ (rocprofv3 --kernel-trace --stats of bench.py, same command as the bench)
set -o pipefail
CFG=${1:-C1}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$CFG
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$CFG -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs --no-f32-mfma-leg > $R/gpurun_out/prof_$CFG.json 2> $R/gpurun_out/prof_$CFG.err; echo "prof exit=$?" >> $R/gpurun_out/prof_$CFG.err
tail -2 $R/gpurun_out/prof_$CFG.err
find $R/gpurun_out/prof_$CFG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof_${CFG}_kernel_stats.csv
rm -rf $R/gpurun_out/prof_$CFG      # (the kernel trace is > 64 MiB; the stats CSV was copied above)
head -25 $R/gpurun_out/prof_${CFG}_kernel_stats.csv | cut -c1-170
