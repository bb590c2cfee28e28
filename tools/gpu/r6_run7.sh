#!/bin/bash
# round 6, call 7: the f16x3 operand format end to end — op tests at the fp32-grade bars, the UNet / e2e fixtures in that mode, step probe
O=gpurun_out/r6_7; mkdir -p $O; export TMPDIR=/tmp
rm -f $O/err_log.tsv
ALDM_ERR_LOG=$O/err_log.tsv timeout 1500 python -m pytest tests/test_f16x3_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | tail -40 > $O/tests_f16x3.txt
cat $O/tests_f16x3.txt
ALDM_ERR_LOG=$O/err_log.tsv timeout 2400 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "f16x3" 2>&1 | tail -40 > $O/tests_model_f16x3.txt
cat $O/tests_model_f16x3.txt
{
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/bf16x6: /'
ALDM_MMA=f16x3 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error\|error" | sed 's/^/f16x3: /'
ALDM_MMA=bf16x3 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/bf16x3: /'
} > $O/step_modes.txt 2>&1; cat $O/step_modes.txt
