#!/bin/bash
# round 3, call 27: the model parity tests in the strict product mode (ALDM_MMA=bf16x6) and on the fp32 MFMA (ALDM_MMA=f32)
mkdir -p gpurun_out/r3
for M in bf16x6 f32; do
rm -f gpurun_out/parity_report.txt
ALDM_MMA=$M timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "unet_ or vae or hifigan or e2e_5step or e2e_200step or masked or ancestral" 2>&1 | tail -2
cp gpurun_out/parity_report.txt gpurun_out/r3/parity_report_$M.txt
done
