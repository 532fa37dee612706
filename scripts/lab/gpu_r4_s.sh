# C2 step by (kernel, grid): rocprofv3 kernel trace of the bench command, aggregated on the box (the raw trace is > 64 MiB)
set -o pipefail
R=$GRAFT_REPO_ROOT
CFG=${1:-C2}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_$CFG
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$CFG -- python3 $R/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --in-order --no-recompute-leg --no-anchor-leg --no-bf16-legs > $R/gpurun_out/trace_$CFG.json 2> $R/gpurun_out/trace_$CFG.err || { tail -5 $R/gpurun_out/trace_$CFG.err; exit 1; }
python3 $R/scripts/trace_by_grid.py $R/gpurun_out/trace_$CFG 3 90 > $R/gpurun_out/trace_${CFG}_by_grid.txt
rm -rf $R/gpurun_out/trace_$CFG
head -70 $R/gpurun_out/trace_${CFG}_by_grid.txt | cut -c1-200
