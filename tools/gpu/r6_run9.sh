#!/bin/bash
# round 6, call 9: f16x3 through the MLP (GEGLU output as an fp16 image): tests, step A/B
O=gpurun_out/r6_9; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_f16x3_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^        \|^    def\|^$" | tail -40 > $O/tests_f16x3.txt
cat $O/tests_f16x3.txt
timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "f16x3 and not bf16x3 and (unet or 5step)" 2>&1 | tail -5
{
for i in 1 2; do
ALDM_MMA=f16x3 ALDM_F16_FF_OUT=0 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed 's/^/f16x3, FF-out bf16x6: /'
ALDM_MMA=f16x3 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed 's/^/f16x3, FF-out f16x3: /'
done
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/bf16x6: /'
} > $O/step_ab_f16_ffout.txt 2>&1; cat $O/step_ab_f16_ffout.txt
