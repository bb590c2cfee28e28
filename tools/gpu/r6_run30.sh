#!/bin/bash
# round 6, call 30: switches that were chosen on isolated launches (rounds 2-3), re-checked INSIDE the replayed bf16x6 step on the final tree:
# LayerNorm rows per wave, loader-wave blocks per CU, the one-launch GroupNorm's size limit, attention queries per wave
O=gpurun_out/r6_30; mkdir -p $O; export TMPDIR=/tmp
run() { env "$@" ALDM_MMA=bf16x6 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed "s/^/$*: /"; }
{
run DEFAULTS=1
run ALDM_LN_R=2
run ALDM_LN_R=4
run ALDM_LW_BPC=1
run ALDM_GN_FUSED_MAX=262144
run ALDM_GN_FUSED_MAX=4194304
run DEFAULTS=1
run ALDM_LN_R=2
} > $O/step_ab_switches_bf16x6.txt 2>&1; cat $O/step_ab_switches_bf16x6.txt
