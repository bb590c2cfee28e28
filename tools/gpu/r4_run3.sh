#!/bin/bash
# round 4: where a stage of the operand-stationary kernel spends its time (ablation builds; bf16x6)
mkdir -p gpurun_out/r4/run3
rm -f gpurun_out/r4/run3/os_ablate.txt
for v in base nodma nomfma noepi nostore nordfrag nomfma_noepi mfma_only; do
  ALDM_LIB_PATH=tools/gpu/libaldm_os_$v.so timeout 300 python tools/os_ablate.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r4/run3/os_ablate.txt
done
