#!/bin/bash
# round 6, call 28: in-step tuning (tools/instep_autotune.py) of the 3-part table in the OTHER configurations' steps — large, 48 k, speech — each
# starting from the table the previous one left; then a same-box A/B of every configuration's step against the committed tables (bf16x6)
O=gpurun_out/r6_28; mkdir -p $O /tmp/tab_new; export TMPDIR=/tmp
cp audioldm2_amd/tuning/mi355x_igemm_dma.json audioldm2_amd/tuning/mi355x_igemm_dma_bf16x3.json /tmp/tab_new/
for MODEL in audioldm2-full-large-1150k audioldm_48k audioldm2-speech-gigaspeech; do
INSTEP_BUDGET_S=${INSTEP_BUDGET_S:-420} ALDM_TUNING_DIR=/tmp/tab_new ALDM_MMA=bf16x6 timeout 900 python tools/instep_autotune.py $O/instep_bf16x6_$MODEL.json $MODEL 40 2>&1 | grep -v amdgpu.ids > $O/instep_autotune_bf16x6_$MODEL.txt
grep -v "kept" $O/instep_autotune_bf16x6_$MODEL.txt | tail -12
python - $MODEL <<'PY'
import json, sys
res = "gpurun_out/r6_28/instep_bf16x6_%s.json" % sys.argv[1]
try:
    ch = json.load(open(res))["changed"]
except OSError:
    ch = {}
t = json.load(open("/tmp/tab_new/mi355x_igemm_dma.json"))
for k, v in ch.items():
    t["entries"][k] = list(v[:4]) + [0, 0]
json.dump(t, open("/tmp/tab_new/mi355x_igemm_dma.json", "w"), indent=0, sort_keys=True)
PY
done
cp /tmp/tab_new/mi355x_igemm_dma.json $O/mi355x_igemm_dma.json
{
for i in 1 2; do
for MODEL in audioldm2-full audioldm2-full-large-1150k audioldm_48k audioldm2-speech-gigaspeech; do
ALDM_MMA=bf16x6 timeout 600 python tools/step_probe.py $MODEL 2 2>&1 | grep "unet step\|Error" | sed "s/^/$MODEL committed table: /"
ALDM_MMA=bf16x6 ALDM_TUNING_DIR=/tmp/tab_new timeout 600 python tools/step_probe.py $MODEL 2 2>&1 | grep "unet step\|Error" | sed "s/^/$MODEL in-step tuned in its own step: /"
done
done
} > $O/step_ab_instep_others.txt 2>&1; cat $O/step_ab_instep_others.txt
