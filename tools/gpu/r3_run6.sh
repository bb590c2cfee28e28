#!/bin/bash
# round 3, call 6: same-box A/B of the round-2 tuned DMA tables vs the round-3 ones (graph-timed, loader-wave / persistent candidates)
mkdir -p gpurun_out/r3
for i in 1 2; do
ALDM_TUNING_DIR=tools/gpu/tuning_r02 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/r02 tables x3: /'
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/r03 tables x3: /'
done > gpurun_out/r3/step_ab_tables.txt 2>&1
ALDM_MMA=bf16x6 ALDM_TUNING_DIR=tools/gpu/tuning_r02 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/r02 tables x6: /' >> gpurun_out/r3/step_ab_tables.txt
ALDM_MMA=bf16x6 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/r03 tables x6: /' >> gpurun_out/r3/step_ab_tables.txt
cat gpurun_out/r3/step_ab_tables.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "unet or e2e_5step" 2>&1 | tail -3
