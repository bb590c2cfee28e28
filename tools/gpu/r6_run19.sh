#!/bin/bash
# round 6, call 19: cheap A/Bs of existing switches in the f16x3 mode (queries per wave of the attention, CFG halves as graph branches)
O=gpurun_out/r6_19; mkdir -p $O; export TMPDIR=/tmp
{
for i in 1 2; do
ALDM_MMA=f16x3 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed 's/^/f16x3 default: /'
ALDM_MMA=f16x3 ALDM_ATTN_QT=1 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed 's/^/f16x3 ALDM_ATTN_QT=1: /'
ALDM_MMA=f16x3 ALDM_CFG_STREAMS=1 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed 's/^/f16x3 ALDM_CFG_STREAMS=1: /'
done
} > $O/step_ab_switches_f16x3.txt 2>&1; cat $O/step_ab_switches_f16x3.txt
