#!/bin/bash
# round 3, call 5: re-tune the DMA-fed GEMMs (graph-timed; candidates now include the loader-wave and persistent forms), bf16x3 then bf16x6
mkdir -p gpurun_out/r3
ALDM_MMA=bf16x3 timeout 1500 python tools/dma_autotune.py gpurun_out/r3/mi355x_igemm_dma_bf16x3.json audioldm2-full > gpurun_out/r3/dma_autotune_bf16x3.txt 2>&1; echo "tune x3 rc=$?"; tail -4 gpurun_out/r3/dma_autotune_bf16x3.txt
ALDM_MMA=bf16x6 timeout 1500 python tools/dma_autotune.py gpurun_out/r3/mi355x_igemm_dma.json audioldm2-full > gpurun_out/r3/dma_autotune_bf16x6.txt 2>&1; echo "tune x6 rc=$?"; tail -4 gpurun_out/r3/dma_autotune_bf16x6.txt
