#!/bin/bash
# round 4, call 5: operand-stationary kernel v2 — tests; re-tune of the geometries it can run (both modes, all four configs),
# merged into the round-3 tables; same-box step A/B round-3 tables vs the new ones
mkdir -p gpurun_out/r4/run5
O=gpurun_out/r4/run5
timeout 900 python -m pytest tests/test_dma_gpu.py -q -m gpu -x -k "dma_os" 2>&1 | tail -5
cd audioldm2_amd/csrc && cd ../..
ALDM_MMA=bf16x6 DMA_TUNE_ONLY_OS=1 DMA_TUNE_MIN_COUNT=4 DMA_TUNE_MERGE=audioldm2_amd/tuning/mi355x_igemm_dma.json timeout 1500 python tools/dma_autotune.py $O/mi355x_igemm_dma.json audioldm2-full audioldm_48k audioldm2-speech-gigaspeech audioldm2-full-large-1150k > $O/dma_autotune_os_bf16x6.txt 2>&1; echo "tune x6 rc=$?"; tail -4 $O/dma_autotune_os_bf16x6.txt
ALDM_MMA=bf16x3 DMA_TUNE_ONLY_OS=1 DMA_TUNE_MIN_COUNT=4 DMA_TUNE_MERGE=audioldm2_amd/tuning/mi355x_igemm_dma_bf16x3.json timeout 1500 python tools/dma_autotune.py $O/mi355x_igemm_dma_bf16x3.json audioldm2-full audioldm_48k audioldm2-speech-gigaspeech audioldm2-full-large-1150k > $O/dma_autotune_os_bf16x3.txt 2>&1; echo "tune x3 rc=$?"; tail -4 $O/dma_autotune_os_bf16x3.txt
mkdir -p /tmp/newtab && cp $O/mi355x_igemm_dma.json $O/mi355x_igemm_dma_bf16x3.json /tmp/newtab/
for i in 1 2; do
ALDM_MMA=bf16x6 ALDM_TUNING_DIR=tools/gpu/tuning_r03 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/r03 tables x6: /'
ALDM_MMA=bf16x6 ALDM_TUNING_DIR=/tmp/newtab timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/r04 tables (OS) x6: /'
done > $O/step_ab_os.txt 2>&1
ALDM_MMA=bf16x3 ALDM_TUNING_DIR=tools/gpu/tuning_r03 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/r03 tables x3: /' >> $O/step_ab_os.txt
ALDM_MMA=bf16x3 ALDM_TUNING_DIR=/tmp/newtab timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/r04 tables (OS) x3: /' >> $O/step_ab_os.txt
cat $O/step_ab_os.txt
