#!/bin/bash
# round 3, call 13: uncond / cond halves as two concurrent graph branches (ALDM_CFG_STREAMS=1), re-measured with this round's kernels
mkdir -p gpurun_out/r3
for i in 1 2; do
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/one 16-sample pass:        /'
ALDM_CFG_STREAMS=1 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/two 8-sample branches:     /'
done | tee gpurun_out/r3/step_ab_cfg_streams.txt
