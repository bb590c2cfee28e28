#!/bin/bash
# round 6, call 23: the 3-part (bf16x6) table — audioldm2-full's entries date from round 5, the other configurations' from round 4: full re-tune
# of all four on the current kernels + halo pass, then same-box A/Bs (step probe per configuration, shipped vs re-tuned)
O=gpurun_out/r6_23; mkdir -p $O /tmp/newtab3; export TMPDIR=/tmp
ALDM_MMA=bf16x6 DMA_TUNE_MIN_COUNT=2 timeout 3000 python tools/dma_autotune.py $O/dma3_all.json audioldm2-full audioldm_48k audioldm2-full-large-1150k audioldm2-speech-gigaspeech > $O/dma_autotune_bf16x6_all.txt 2>&1; echo "tune rc=$?"; tail -2 $O/dma_autotune_bf16x6_all.txt
python - <<'PY'
import json
cur = json.load(open("audioldm2_amd/tuning/mi355x_igemm_dma.json"))
new = json.load(open("gpurun_out/r6_23/dma3_all.json"))
ent = dict(cur["entries"]); ent.update(new["entries"])
json.dump({"device": "MI355X", "kernel": "igemm_dma_kernel", "parts": 3, "entries": ent}, open("gpurun_out/r6_23/dma3_merged.json", "w"), indent=0, sort_keys=True)
print("merged entries:", len(ent), "re-tuned:", len(new["entries"]))
PY
ALDM_MMA=bf16x6 DMA_TUNE_ONLY_HALO=1 DMA_TUNE_MERGE=$O/dma3_merged.json DMA_TUNE_MIN_COUNT=2 timeout 1500 python tools/dma_autotune.py $O/dma3_merged_halo.json audioldm2-full audioldm_48k audioldm2-full-large-1150k audioldm2-speech-gigaspeech > $O/halo_autotune_bf16x6_all.txt 2>&1; echo "halo tune rc=$?"; tail -1 $O/halo_autotune_bf16x6_all.txt
cp $O/dma3_merged_halo.json /tmp/newtab3/mi355x_igemm_dma.json
{
for i in 1 2; do
for M in audioldm2-full audioldm_48k audioldm2-full-large-1150k audioldm2-speech-gigaspeech; do
timeout 600 python tools/step_probe.py $M 2 2>&1 | grep "unet step\|Error" | sed "s/^/$M shipped 3-part table: /"
ALDM_TUNING_DIR=/tmp/newtab3 timeout 600 python tools/step_probe.py $M 2 2>&1 | grep "unet step\|Error" | sed "s/^/$M re-tuned 3-part table: /"
done
done
} > $O/step_ab_tables_3part.txt 2>&1; cat $O/step_ab_tables_3part.txt
