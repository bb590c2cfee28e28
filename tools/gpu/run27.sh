set -x
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -k "two_rank" > gpurun_out/model_test_shard.log 2>&1; echo "rc=$?"; tail -15 gpurun_out/model_test_shard.log
ALDM_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 1 --warmup 0 --ddim-steps 10 --batch 4 --no-cpu-baseline --no-roofline > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err; echo "bench rc=$?"; tail -3 gpurun_out/bench_2rank_gloo.err; cat gpurun_out/bench_2rank_gloo.json | cut -c1-400
