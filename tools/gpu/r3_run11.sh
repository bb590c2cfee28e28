#!/bin/bash
# round 3, call 11: pre-split attention operands — parity (bitwise vs the fp32 K/V path), model parity, step A/B; bench `configs`
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_dma_gpu.py -x -q -k "qkv_epilogue" > gpurun_out/r3/presplit_tests.log 2>&1; echo "presplit tests rc=$?"; tail -12 gpurun_out/r3/presplit_tests.log | cut -c1-300
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "unet or e2e_5step" 2>&1 | tail -3
for i in 1 2; do
ALDM_ATTN_PRESPLIT=0 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/fp32 K,V split in the attention loop: /'
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/pre-split K, V^T from the qkv GEMM:  /'
done | tee gpurun_out/r3/step_ab_presplit.txt
