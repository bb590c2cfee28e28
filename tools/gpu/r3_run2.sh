#!/bin/bash
# round 3, call 2: loader-wave DMA GEMM (igemm_dma_lw.h) vs igemm_dma_kernel: bitwise check + graph-timed A/B, 2 and 1 blocks per CU
mkdir -p gpurun_out/r3
WS_PROBE=lw timeout 900 python tools/ws_probe.py bf16x3 > gpurun_out/r3/lw_probe_bf16x3.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r3/lw_probe_bf16x3.txt | cut -c1-600
ALDM_LW_BPC=1 WS_PROBE=lw timeout 900 python tools/ws_probe.py bf16x3 > gpurun_out/r3/lw_probe_bf16x3_bpc1.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r3/lw_probe_bf16x3_bpc1.txt | cut -c1-600
