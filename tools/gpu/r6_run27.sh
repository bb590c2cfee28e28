#!/bin/bash
# round 6, call 27: in-step tuning, second pass over ALL of the bf16x6 step's geometries (from the table pass 1 produced), then the 2-part
# geometries of the f16x3 step; same-box A/B of each against the tables of the end-of-round records (tools/gpu/tuning_r06_records/)
O=gpurun_out/r6_27; mkdir -p $O /tmp/tab_new; export TMPDIR=/tmp
INSTEP_BUDGET_S=900 ALDM_MMA=bf16x6 timeout 1800 python tools/instep_autotune.py $O/instep_bf16x6_pass2.json audioldm2-full 60 2>&1 | grep -v amdgpu.ids > $O/instep_autotune_bf16x6_pass2.txt; grep -v "kept" $O/instep_autotune_bf16x6_pass2.txt | tail -20
INSTEP_BUDGET_S=900 INSTEP_ONLY=dma2 ALDM_MMA=f16x3 timeout 1800 python tools/instep_autotune.py $O/instep_f16x3.json audioldm2-full 60 2>&1 | grep -v amdgpu.ids > $O/instep_autotune_f16x3.txt; grep -v "kept" $O/instep_autotune_f16x3.txt | tail -20
python - <<'PY'
import json, shutil
for name, res in (("mi355x_igemm_dma.json", "gpurun_out/r6_27/instep_bf16x6_pass2.json"), ("mi355x_igemm_dma_bf16x3.json", "gpurun_out/r6_27/instep_f16x3.json")):
    t = json.load(open("audioldm2_amd/tuning/" + name))
    for k, v in json.load(open(res))["changed"].items():
        t["entries"][k] = list(v[:4]) + [0, 0]
    json.dump(t, open("/tmp/tab_new/" + name, "w"), indent=0, sort_keys=True)
    shutil.copy("/tmp/tab_new/" + name, "gpurun_out/r6_27/" + name)
PY
{
for i in 1 2; do
for MODE in bf16x6 f16x3; do
ALDM_MMA=$MODE ALDM_TUNING_DIR=tools/gpu/tuning_r06_records timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed "s/^/$MODE tables of the end-of-round records: /"
ALDM_MMA=$MODE ALDM_TUNING_DIR=/tmp/tab_new timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed "s/^/$MODE in-step tuned tables: /"
done
done
} > $O/step_ab_instep.txt 2>&1; cat $O/step_ab_instep.txt
