#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_dma_gpu.py -x -q -m gpu 2>&1 | tail -3
ALDM_MMA=bf16x3 timeout 900 python tools/dma_autotune.py gpurun_out/r2/mi355x_igemm_dma_bf16x3.json audioldm2-full > gpurun_out/r2/dma_autotune_x3_v2.txt 2>&1; echo "autotune rc=$?"; grep -c "st6" gpurun_out/r2/dma_autotune_x3_v2.txt; tail -3 gpurun_out/r2/dma_autotune_x3_v2.txt
