#!/bin/bash
# round 3, call 1: the persistent wave-specialised DMA GEMM — parity tests (bitwise vs igemm_dma_kernel), then same-box timing
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_dma_gpu.py -x -q -k "ws" > gpurun_out/r3/ws_tests.log 2>&1; echo "ws tests rc=$?"; tail -5 gpurun_out/r3/ws_tests.log | cut -c1-300
timeout 900 python tools/ws_probe.py bf16x3 > gpurun_out/r3/ws_probe_bf16x3.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r3/ws_probe_bf16x3.txt | cut -c1-400
