#!/bin/bash
# round 6, call 16: per-step breakdown (tools/trace_step_breakdown.py) of the replayed DDIM steps in both fp32-grade modes; the new saturation test
O=gpurun_out/r6_16; mkdir -p $O; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_f16x3_gpu.py -q -m gpu -p no:cacheprovider -k "saturate" 2>&1 | grep -B30 "short test summary" | grep "^E\|assert" | head -20
for MODE in bf16x6 f16x3; do
cd /tmp
rm -rf /tmp/prof_$MODE /tmp/kt_$MODE
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$MODE -o fin --output-format csv -- python $R/bench.py --mma $MODE --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-f16x3 --no-configs --no-conditioners --no-api-default < /dev/null > /dev/null 2>&1
cd $R
mkdir -p /tmp/kt_$MODE && cp $(find /tmp/prof_$MODE -name "*kernel_trace.csv" | head -1) /tmp/kt_$MODE/
python tools/trace_step_breakdown.py /tmp/kt_$MODE 45 > $O/step_breakdown_$MODE.txt 2>&1; head -70 $O/step_breakdown_$MODE.txt
done
