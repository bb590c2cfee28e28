#!/bin/bash
mkdir -p gpurun_out/r2
for r in 0 1 2 8; do echo "== ALDM_LN_R=$r"; ALDM_LN_R=$r python tools/ln_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2/ln_bench.txt
