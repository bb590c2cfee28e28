#!/bin/bash
# round 6, call 24: the re-tuned 2-part table in the step of the OTHER three configurations (f16x3), against the table it replaces
# (tools/gpu/tuning_r05_2part/: the round-4/5 entries)
O=gpurun_out/r6_24; mkdir -p $O; export TMPDIR=/tmp
{
for i in 1 2; do
for M in audioldm_48k audioldm2-full-large-1150k audioldm2-speech-gigaspeech; do
ALDM_MMA=f16x3 ALDM_TUNING_DIR=tools/gpu/tuning_r05_2part timeout 600 python tools/step_probe.py $M 2 2>&1 | grep "unet step\|Error" | sed "s/^/$M f16x3, replaced 2-part table: /"
ALDM_MMA=f16x3 timeout 600 python tools/step_probe.py $M 2 2>&1 | grep "unet step\|Error" | sed "s/^/$M f16x3, re-tuned 2-part table: /"
done
done
} > $O/step_ab_tables_2part_others.txt 2>&1; cat $O/step_ab_tables_2part_others.txt
