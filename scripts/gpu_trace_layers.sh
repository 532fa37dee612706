# Per-layer HBM traffic of one training step of a bench.py configuration:
#   bash scripts/gpu_trace_layers.sh C2   -> gpurun_out/<cfg>_traffic_by_layer.txt
# Three rocprofv3 passes of scripts/step_trace.py (FETCH_SIZE; WRITE_SIZE; plain kernel trace), joined call by call by
# scripts/trace_traffic.py.  Counters are collected in their own passes with --kernel-trace only.
set -o pipefail
CFG=${1:-C2}
R=$GRAFT_REPO_ROOT
L=$(echo $CFG | tr 'A-Z' 'a-z')
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE TIME; do
  D=$R/gpurun_out/trace_${CFG}_$c
  rm -rf $D
  if [ $c = TIME ]; then PMC=""; else PMC="--pmc $c"; fi
  timeout -k 10 420 rocprofv3 $PMC --kernel-trace --output-format csv -d $D -- python3 $R/scripts/step_trace.py --config $CFG --out $D.calls.json > $D.log 2>&1
  rc=$?; echo "trace $c exit=$rc" >> $D.log; tail -2 $D.log
  [ $rc -eq 0 ] || exit $rc
done
cd $R && python3 scripts/trace_traffic.py gpurun_out/trace_${CFG}_FETCH_SIZE.calls.json gpurun_out/trace_${CFG}_FETCH_SIZE gpurun_out/trace_${CFG}_WRITE_SIZE gpurun_out/trace_${CFG}_TIME > gpurun_out/${L}_traffic_by_layer.txt || exit 1
# the raw per-dispatch CSVs are tens of MB each (gpurun copies back at most 64 MiB): keep the table, the call log and the logs
rm -rf gpurun_out/trace_${CFG}_FETCH_SIZE gpurun_out/trace_${CFG}_WRITE_SIZE gpurun_out/trace_${CFG}_TIME gpurun_out/trace_${CFG}_WRITE_SIZE.calls.json gpurun_out/trace_${CFG}_TIME.calls.json
head -40 gpurun_out/${L}_traffic_by_layer.txt
