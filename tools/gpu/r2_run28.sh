#!/bin/bash
mkdir -p gpurun_out/r2
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_gap -o gap --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
cd $R
python tools/trace_gaps.py /tmp/prof_gap | tee gpurun_out/r2/trace_gaps.txt
