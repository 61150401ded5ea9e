# one GPU session of the round (edited per call; results under gpurun_out/<tag>)
T=${1:-r4w}
R=$PWD; O=$R/gpurun_out/$T; mkdir -p $O
LIDIFF_BENCH_BACKEND=gloo LIDIFF_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 6 --warmup 2 > $O/bench_2ranks_shared.json 2> $O/bench_2ranks_shared.err; echo rc=$?; wc -l $O/bench_2ranks_shared.json; cut -c1-700 $O/bench_2ranks_shared.json; tail -5 $O/bench_2ranks_shared.err | cut -c1-200
LIDIFF_BENCH_BACKEND=gloo LIDIFF_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --pipeline --scans 1 > $O/bench_pipeline_2ranks_shared.json 2> $O/bench_pipeline_2ranks_shared.err; echo rc=$?; wc -l $O/bench_pipeline_2ranks_shared.json; cut -c1-600 $O/bench_pipeline_2ranks_shared.json; tail -3 $O/bench_pipeline_2ranks_shared.err | cut -c1-200
