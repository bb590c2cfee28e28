#!/bin/bash
# round 5, call 9: 16 prompts on one GPU three ways, same box: one process x 16 prompts (32-row passes, geometries the tables do not hold),
# two processes x 8 prompts (call 8's winner), and one process x 8 prompts for scale
O=gpurun_out/r5_9; mkdir -p $O; export TMPDIR=/tmp
Q="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-configs --no-conditioners --no-api-default --no-replicas"
P='import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(sys.argv[1], d["value"], "audio-s/s", d["ms_per_step"], "ms per job", d.get("per_rank_seconds", ""))'
{
timeout -k 5 600 python3 bench.py --gpus 1 --batch 8 $Q < /dev/null 2>/dev/null | python -c "$P" "1 process x 8 prompts  :"
timeout -k 5 900 python3 bench.py --gpus 1 --batch 16 $Q < /dev/null 2>/dev/null | python -c "$P" "1 process x 16 prompts :"
ALDM_DIST_BACKEND=gloo timeout -k 5 900 python3 bench.py --gpus 2 --batch 8 $Q < /dev/null 2>/dev/null | python -c "$P" "2 processes x 8 prompts:"
ALDM_DIST_BACKEND=gloo timeout -k 5 900 python3 bench.py --gpus 3 --batch 8 $Q < /dev/null 2>/dev/null | python -c "$P" "3 processes x 8 prompts:"
} 2>&1 | tee $O/sixteen_prompts_one_gpu.txt
