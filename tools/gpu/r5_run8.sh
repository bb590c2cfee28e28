#!/bin/bash
# round 5, call 8: does ONE GPU run 8 prompts faster as two concurrent 4-prompt replicas (two processes, two HW queues) than as one
# 8-prompt job?  (call 7: two 8-prompt ranks sharing the GPU delivered 19.8 audio-s/s against 18.2 for one.)  Same box, same flags.
O=gpurun_out/r5_8; mkdir -p $O; export TMPDIR=/tmp
Q="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-configs --no-conditioners --no-api-default"
for i in 1 2; do
timeout -k 5 600 python3 bench.py --gpus 1 --batch 8 $Q < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('1 process x 8 prompts :', d['value'], 'audio-s/s', d['ms_per_step'], 'ms per job')"
ALDM_DIST_BACKEND=gloo timeout -k 5 600 python3 bench.py --gpus 2 --batch 4 $Q < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('2 processes x 4 prompts:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms per job', d['per_rank_seconds'])"
ALDM_DIST_BACKEND=gloo timeout -k 5 600 python3 bench.py --gpus 2 --batch 8 $Q < /dev/null 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print('2 processes x 8 prompts:', d['value'], 'audio-s/s', d['ms_per_step'], 'ms per job', d['per_rank_seconds'])"
done 2>&1 | tee $O/replicas_one_gpu.txt
