set -o pipefail
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
# N=2 rehearsal of the DP path on one GPU (gloo, both ranks on cuda:0), C0 shapes
EDRL_DIST_BACKEND=gloo EDRL_DEVICE=0 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --config C0 --steps 3 --warmup 2 > gpurun_out/bench_dp2_gloo.json 2> gpurun_out/bench_dp2_gloo.err; echo "dp2 exit=$?" >> gpurun_out/bench_dp2_gloo.err
cat gpurun_out/bench_dp2_gloo.json; tail -3 gpurun_out/bench_dp2_gloo.err
timeout -k 10 300 python bench.py --config C0 --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/bench_c0.json 2> gpurun_out/bench_c0.err; echo "c0 exit=$?" >> gpurun_out/bench_c0.err
cat gpurun_out/bench_c0.json
timeout -k 10 300 python scripts/conv_layer_bench.py 1024 > gpurun_out/layers_1024.log 2>&1; echo "layers exit=$?" >> gpurun_out/layers_1024.log
cat gpurun_out/layers_1024.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof2_b8 -- python3 $R/bench.py --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timing > $R/gpurun_out/prof2_b8.log 2>&1; echo "prof exit=$?" >> $R/gpurun_out/prof2_b8.log
head -22 $R/gpurun_out/prof2_b8/*/*kernel_stats.csv | cut -c1-150
