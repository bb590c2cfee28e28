#!/bin/bash
# round 6, call 11: per-grid kernel trace of a 10-step job in the headline mode (bf16x6) and in f16x3 (K / V^T fp16 images, FF-out fp16) — what is left
O=gpurun_out/r6_11; mkdir -p $O; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
Q="--steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-f16x3 --no-configs --no-conditioners --no-api-default"
for MODE in bf16x6 f16x3; do
cd /tmp
rm -rf /tmp/prof_$MODE /tmp/kt_$MODE
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$MODE -o fin --output-format csv -- python $R/bench.py --mma $MODE $Q < /dev/null > /dev/null 2>&1
cd $R
cp $(find /tmp/prof_$MODE -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$MODE.csv
mkdir -p /tmp/kt_$MODE && cp $(find /tmp/prof_$MODE -name "*kernel_trace.csv" | head -1) /tmp/kt_$MODE/ && python tools/trace_by_grid.py /tmp/kt_$MODE 100 > $O/trace_by_grid_$MODE.txt 2>&1
head -14 $O/kernel_stats_$MODE.csv | cut -c1-170
done
