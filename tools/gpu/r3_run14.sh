#!/bin/bash
# round 3, call 14: core clock and package power while the UNet step graph replays (is the step power limited?)
mkdir -p gpurun_out/r3
( for i in $(seq 1 120); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "sclk|Power|GPU use" | tr '\n' ' ' ; echo; sleep 0.5; done ) > gpurun_out/r3/smi_during_step.txt 2>&1 &
SMI=$!
timeout 600 python tools/step_probe.py audioldm2-full 6 2>&1 | grep "unet step"
kill $SMI 2>/dev/null
sort gpurun_out/r3/smi_during_step.txt | uniq -c | sort -rn | head -25
