#!/bin/bash
# round 3, call 16: tune the DMA-fed GEMMs of the other three configurations (default mode; UNet geometries only), merged into the table
mkdir -p gpurun_out/r3
ALDM_MMA=bf16x3 DMA_TUNE_MERGE=audioldm2_amd/tuning/mi355x_igemm_dma_bf16x3.json DMA_TUNE_MIN_COUNT=4 timeout 2400 python tools/dma_autotune.py gpurun_out/r3/mi355x_igemm_dma_bf16x3_all.json audioldm_48k audioldm2-speech-gigaspeech audioldm2-full-large-1150k > gpurun_out/r3/dma_autotune_bf16x3_other.txt 2>&1; echo "tune rc=$?"; tail -3 gpurun_out/r3/dma_autotune_bf16x3_other.txt
