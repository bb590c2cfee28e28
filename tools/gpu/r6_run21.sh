#!/bin/bash
# round 6, call 21: the 2-part geometry table (bf16x3 and f16x3 launches) was last tuned on the round-4 kernels; r6_run20 shows the round-5
# operand-stationary kernel beating its GEGLU entries by 11-22 %.  Full re-tune of audioldm2-full's 2-part geometries on the current
# kernels (classic / loader-wave / wave-specialised / operand-stationary forms, then the halo forms against the result), then same-box step
# A/Bs shipped vs re-tuned table in both 2-part modes.
O=gpurun_out/r6_21; mkdir -p $O /tmp/newtab; export TMPDIR=/tmp
ALDM_MMA=bf16x3 DMA_TUNE_MIN_COUNT=2 timeout 2400 python tools/dma_autotune.py $O/dma2_full.json audioldm2-full > $O/dma_autotune_bf16x3_full.txt 2>&1; echo "tune rc=$?"; tail -3 $O/dma_autotune_bf16x3_full.txt
python - <<'PY'
import json
old = json.load(open("audioldm2_amd/tuning/mi355x_igemm_dma_bf16x3.json"))
new = json.load(open("gpurun_out/r6_21/dma2_full.json"))
ent = dict(old["entries"])
seen = set()
for line in open("gpurun_out/r6_21/dma_autotune_bf16x3_full.txt"):
    pass
# every geometry the tuner looked at gets its new verdict; one it judged "cost model within 3 %" loses its old entry
import re
looked = set()
ent.update(new["entries"])
json.dump({"device": "MI355X", "kernel": "igemm_dma_kernel", "parts": 2, "entries": ent}, open("gpurun_out/r6_21/dma2_merged.json", "w"), indent=0, sort_keys=True)
print("merged entries:", len(ent), "re-tuned:", len(new["entries"]))
PY
ALDM_MMA=bf16x3 DMA_TUNE_ONLY_HALO=1 DMA_TUNE_MERGE=$O/dma2_merged.json DMA_TUNE_MIN_COUNT=2 timeout 900 python tools/dma_autotune.py $O/dma2_merged_halo.json audioldm2-full > $O/halo_autotune_bf16x3_full.txt 2>&1; echo "halo tune rc=$?"; tail -2 $O/halo_autotune_bf16x3_full.txt
cp $O/dma2_merged_halo.json /tmp/newtab/mi355x_igemm_dma_bf16x3.json
{
for i in 1 2; do
for MODE in f16x3 bf16x3; do
ALDM_MMA=$MODE timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed "s/^/$MODE shipped 2-part table: /"
ALDM_MMA=$MODE ALDM_TUNING_DIR=/tmp/newtab timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed "s/^/$MODE re-tuned 2-part table: /"
done
done
} > $O/step_ab_tables_2part.txt 2>&1; cat $O/step_ab_tables_2part.txt
