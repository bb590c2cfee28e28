#!/bin/bash
# round 6, call 22: the other three configurations' 2-part geometries re-tuned on the current kernels (merged into the table call 21 produced),
# halo pass, model / config parity tests in the two 2-part modes on the new table, then every configuration's job in f16x3 (value + UNet step)
O=gpurun_out/r6_22; mkdir -p $O; export TMPDIR=/tmp
ALDM_MMA=bf16x3 DMA_TUNE_MIN_COUNT=2 timeout 2400 python tools/dma_autotune.py $O/dma2_others.json audioldm_48k audioldm2-full-large-1150k audioldm2-speech-gigaspeech > $O/dma_autotune_bf16x3_others.txt 2>&1; echo "tune rc=$?"; tail -2 $O/dma_autotune_bf16x3_others.txt
python - <<'PY'
import json
cur = json.load(open("audioldm2_amd/tuning/mi355x_igemm_dma_bf16x3.json"))
new = json.load(open("gpurun_out/r6_22/dma2_others.json"))
full = json.load(open("gpurun_out/r6_21/dma2_full.json")) if __import__("os").path.exists("gpurun_out/r6_21/dma2_full.json") else {"entries": {}}
ent = dict(cur["entries"])
for k, v in new["entries"].items():
    if k not in full["entries"]:      # (audioldm2-full's verdicts of call 21 stand: same kernels, its own box)
        ent[k] = v
json.dump({"device": "MI355X", "kernel": "igemm_dma_kernel", "parts": 2, "entries": ent}, open("gpurun_out/r6_22/dma2_merged.json", "w"), indent=0, sort_keys=True)
print("merged entries:", len(ent), "re-tuned for the other configs:", len(new["entries"]))
PY
ALDM_MMA=bf16x3 DMA_TUNE_ONLY_HALO=1 DMA_TUNE_MERGE=$O/dma2_merged.json DMA_TUNE_MIN_COUNT=2 timeout 1200 python tools/dma_autotune.py $O/dma2_merged_halo.json audioldm_48k audioldm2-full-large-1150k audioldm2-speech-gigaspeech > $O/halo_autotune_bf16x3_others.txt 2>&1; echo "halo tune rc=$?"; tail -1 $O/halo_autotune_bf16x3_others.txt
cp $O/dma2_merged_halo.json audioldm2_amd/tuning/mi355x_igemm_dma_bf16x3.json
rm -f $O/err_log.tsv
ALDM_ERR_LOG=$O/err_log.tsv timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_parity_configs_gpu.py tests/test_f16x3_gpu.py -q -m gpu -p no:cacheprovider -k "bf16x3 or f16x3" 2>&1 | tail -4 | tee $O/tests_2part_modes.txt
Q="--steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-fast --no-f16x3 --no-configs --no-conditioners --no-api-default --no-replicas"
for M in audioldm2-full audioldm_48k audioldm2-full-large-1150k audioldm2-speech-gigaspeech; do
for MODE in bf16x6 f16x3; do
timeout 600 python bench.py --model $M --mma $MODE $Q 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.readline())
print('$M $MODE: %.2f audio-s/s, UNet step %.2f ms' % (j['value'], j.get('unet_step_ms', float('nan'))))"
done
done | tee $O/configs_f16x3.txt
