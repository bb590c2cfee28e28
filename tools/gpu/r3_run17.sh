#!/bin/bash
# round 3, call 17: tune every remaining DMA-fed geometry of the four configurations (VAE / vocoder launches included), both modes
mkdir -p gpurun_out/r3
ALDM_MMA=bf16x3 DMA_TUNE_MERGE=audioldm2_amd/tuning/mi355x_igemm_dma_bf16x3.json timeout 2400 python tools/dma_autotune.py gpurun_out/r3/mi355x_igemm_dma_bf16x3_all2.json audioldm2-full audioldm_48k audioldm2-speech-gigaspeech audioldm2-full-large-1150k > gpurun_out/r3/dma_autotune_bf16x3_rest.txt 2>&1; echo "tune x3 rc=$?"; tail -2 gpurun_out/r3/dma_autotune_bf16x3_rest.txt
ALDM_MMA=bf16x6 DMA_TUNE_MERGE=audioldm2_amd/tuning/mi355x_igemm_dma.json timeout 2400 python tools/dma_autotune.py gpurun_out/r3/mi355x_igemm_dma_all2.json audioldm2-full audioldm_48k audioldm2-speech-gigaspeech audioldm2-full-large-1150k > gpurun_out/r3/dma_autotune_bf16x6_rest.txt 2>&1; echo "tune x6 rc=$?"; tail -2 gpurun_out/r3/dma_autotune_bf16x6_rest.txt
