#!/bin/bash
# round 6, call 4: re-tune of the geometries the halo-patch kernel can run (all four configs, bf16x6; audioldm2-full also bf16x3)
# against their current table entries, then the same-box step A/B shipped tables vs halo tables
O=gpurun_out/r6_4; mkdir -p $O/tuning; export TMPDIR=/tmp
T=audioldm2_amd/tuning
DMA_TUNE_ONLY_HALO=1 DMA_TUNE_MERGE=$T/mi355x_igemm_dma.json timeout 2400 python tools/dma_autotune.py $O/tuning/mi355x_igemm_dma.json audioldm2-full audioldm2-full-large-1150k audioldm2-speech-gigaspeech audioldm_48k 2>&1 | grep -v amdgpu.ids > $O/halo_autotune_bf16x6.txt
tail -70 $O/halo_autotune_bf16x6.txt
ALDM_MMA=bf16x3 DMA_TUNE_ONLY_HALO=1 DMA_TUNE_MERGE=$T/mi355x_igemm_dma_bf16x3.json timeout 1200 python tools/dma_autotune.py $O/tuning/mi355x_igemm_dma_bf16x3.json audioldm2-full 2>&1 | grep -v amdgpu.ids > $O/halo_autotune_bf16x3.txt
tail -30 $O/halo_autotune_bf16x3.txt
{
for i in 1 2; do
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/shipped tables: /'
ALDM_TUNING_DIR=$O/tuning timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/halo tables: /'
done
timeout 600 python tools/step_probe.py audioldm2-full-large-1150k 2 2>&1 | grep "unet step" | sed 's/^/large, shipped tables: /'
ALDM_TUNING_DIR=$O/tuning timeout 600 python tools/step_probe.py audioldm2-full-large-1150k 2 2>&1 | grep "unet step" | sed 's/^/large, halo tables: /'
} > $O/step_ab_halo.txt 2>&1; cat $O/step_ab_halo.txt
