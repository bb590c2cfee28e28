#!/bin/bash
# round 3, call 7: fused GroupNorm + split (one launch), bf16x3 stress tests, op suites; step A/B with the fusion off / on
mkdir -p gpurun_out/r3
timeout 1200 python -m pytest tests/test_dma_gpu.py tests/test_ops_gpu.py -x -q > gpurun_out/r3/ops_suites.log 2>&1; echo "op suites rc=$?"; tail -4 gpurun_out/r3/ops_suites.log | cut -c1-300
grep stress gpurun_out/parity_report.txt
for i in 1 2; do
ALDM_GN_SPLIT_FUSED=0 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/gn+split two launches: /'
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/gn+split fused:        /'
done | tee gpurun_out/r3/step_ab_gn_fused.txt
