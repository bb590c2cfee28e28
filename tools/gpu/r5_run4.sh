#!/bin/bash
# round 5, call 4: full re-tune of audioldm2-full's DMA-fed geometries (bf16x6) on the round-5 kernels (OS epilogue rewrite, shared
# CFG prefix => 8-sample shapes at the head of the UNet), then a same-box step A/B shipped table vs re-tuned table
O=gpurun_out/r5_4; mkdir -p $O /tmp/newtab; export TMPDIR=/tmp
ALDM_MMA=bf16x6 DMA_TUNE_MIN_COUNT=2 timeout 1500 python tools/dma_autotune.py $O/mi355x_igemm_dma_full.json audioldm2-full > $O/dma_autotune_bf16x6_full.txt 2>&1; echo "tune rc=$?"; tail -3 $O/dma_autotune_bf16x6_full.txt
python - <<'PY'
import json
old = json.load(open("audioldm2_amd/tuning/mi355x_igemm_dma.json"))
new = json.load(open("gpurun_out/r5_4/mi355x_igemm_dma_full.json"))
tuned_keys = set()
for line in open("gpurun_out/r5_4/dma_autotune_bf16x6_full.txt"):
    pass
ent = dict(old["entries"])
# every geometry the tuner looked at: its new verdict replaces the old entry (absent from `new` = the cost model's choice is within 3 %)
import re
seen = set(new["entries"])
ent.update(new["entries"])
json.dump({"device": "MI355X", "kernel": "igemm_dma_kernel", "parts": 3, "entries": ent}, open("/tmp/newtab/mi355x_igemm_dma.json", "w"), indent=0, sort_keys=True)
json.dump({"device": "MI355X", "kernel": "igemm_dma_kernel", "parts": 3, "entries": ent}, open("gpurun_out/r5_4/mi355x_igemm_dma_merged.json", "w"), indent=0, sort_keys=True)
print("merged entries:", len(ent), "re-tuned:", len(seen))
PY
{
for i in 1 2; do
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/shipped table: /'
ALDM_TUNING_DIR=/tmp/newtab timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/re-tuned table: /'
done
} > $O/step_ab_tables.txt 2>&1; cat $O/step_ab_tables.txt
