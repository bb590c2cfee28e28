#!/bin/bash
# round 3, call 9: sharding with candidates on the GPU; attention abort stress; copyBuffer attribution; 2-rank gloo bench at 200 steps
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "two_rank" > gpurun_out/r3/shard_tests.log 2>&1; echo "shard tests rc=$?"; tail -3 gpurun_out/r3/shard_tests.log | cut -c1-300; grep "2-rank" gpurun_out/parity_report.txt
timeout 900 python tools/attn_repeat.py 200 > gpurun_out/r3/attn_repeat.txt 2>&1; echo "attn repeat rc=$?"; tail -3 gpurun_out/r3/attn_repeat.txt
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_cp -o cp --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe --no-strict > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_copy_attrib.py /tmp/prof_cp 10 > gpurun_out/r3/trace_copy_attrib.txt 2>&1; cat gpurun_out/r3/trace_copy_attrib.txt
ALDM_DIST_BACKEND=gloo timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --batch 8 --no-cpu-baseline --no-roofline --no-step-probe --no-strict > gpurun_out/r3/bench_2rank_one_gpu_gloo.json 2> gpurun_out/r3/bench_2rank.err; echo "2-rank bench rc=$?"; tail -2 gpurun_out/r3/bench_2rank.err | cut -c1-300; cut -c1-700 gpurun_out/r3/bench_2rank_one_gpu_gloo.json
