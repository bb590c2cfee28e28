#!/bin/bash
# round 6, call 13: the f16x3 attention's output as an fp16 image (to_out in three products): tests, model fixtures in the mode, step A/B
O=gpurun_out/r6_13; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_f16x3_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^        \|^    def\|^$" | tail -40 > $O/tests_f16x3.txt
tail -12 $O/tests_f16x3.txt
rm -f $O/err_log.tsv
ALDM_ERR_LOG=$O/err_log.tsv timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "f16x3 and not bf16x3 and (unet or 5step or 200step)" 2>&1 | tail -5
grep -P "\tf16x3\t" $O/err_log.tsv | cut -c1-160
{
for i in 1 2; do
ALDM_MMA=f16x3 ALDM_F16_ATTN_OUT=0 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed 's/^/f16x3, to_out bf16x6: /'
ALDM_MMA=f16x3 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed 's/^/f16x3, to_out f16x3: /'
done
} > $O/step_ab_f16_attn_out.txt 2>&1; cat $O/step_ab_f16_attn_out.txt
