#!/bin/bash
# round 4: clock / power while the GEGLU GEMM of UNet level 1 runs in a loop: classic tiles, operand-stationary, MFMA-only ablation
mkdir -p gpurun_out/r4/run4
O=gpurun_out/r4/run4/os_power.txt
rm -f $O
for w in 64x128 256x128 os3; do timeout 120 python tools/os_power.py bf16x6 $w 3 2>&1 | grep -v amdgpu.ids | tee -a $O; done
ALDM_LIB_PATH=tools/gpu/libaldm_os_mfma_only.so timeout 120 python tools/os_power.py bf16x6 os3 3 2>&1 | grep -v amdgpu.ids | tee -a $O
ALDM_LIB_PATH=tools/gpu/libaldm_os_nomfma.so timeout 120 python tools/os_power.py bf16x6 os3 3 2>&1 | grep -v amdgpu.ids | tee -a $O
ALDM_LIB_PATH=tools/gpu/libaldm_os_noepi.so timeout 120 python tools/os_power.py bf16x6 os3 3 2>&1 | grep -v amdgpu.ids | tee -a $O
for w in 64x128 os4; do timeout 120 python tools/os_power.py bf16x3 $w 3 2>&1 | grep -v amdgpu.ids | tee -a $O; done
