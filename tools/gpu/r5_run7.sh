#!/bin/bash
# round 5, call 7: `python bench.py --gpus 2` THROUGH ITS OWN LAUNCHER on the one GPU of the box (ALDM_DIST_BACKEND=gloo lets two ranks share it;
# with 2 GPUs the same command runs RCCL): torch.distributed.run is spawned by bench.py itself, weights are broadcast, the 16 prompts
# are sharded 8 + 8, rank 0 prints the contract line with per-rank seconds.  Full 200 steps, 1 timed job.
O=gpurun_out/r5_7; mkdir -p $O; export TMPDIR=/tmp
( time ALDM_DIST_BACKEND=gloo timeout -k 5 900 python3 bench.py --gpus 2 --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-configs --no-conditioners --no-api-default < /dev/null ) > $O/bench_2rank_self_launch.json 2> $O/bench_2rank_self_launch.err; echo "rc=$?"
tail -5 $O/bench_2rank_self_launch.err; cut -c1-1500 $O/bench_2rank_self_launch.json
