#!/bin/bash
# round 3, call 3: persistent wave-specialised kernel with TWO blocks per CU (unpadded hand-off buffer) — tests + probe
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_dma_gpu.py -x -q -k "ws" > gpurun_out/r3/ws_tests2.log 2>&1; echo "ws tests rc=$?"; tail -3 gpurun_out/r3/ws_tests2.log | cut -c1-300
WS_PROBE=ws timeout 900 python tools/ws_probe.py bf16x3 > gpurun_out/r3/ws_probe2_bf16x3.txt 2>&1; echo "probe rc=$?"; cat gpurun_out/r3/ws_probe2_bf16x3.txt | cut -c1-600
