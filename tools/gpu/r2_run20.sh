#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "gn or groupnorm or group_norm or layernorm or ln" 2>&1 | tail -3
for m in 0 204800 1048576 4194304; do echo "== ALDM_GN_FUSED_MAX=$m"; ALDM_GN_FUSED_MAX=$m python tools/gn_bench.py 2>&1 | grep -v amdgpu.ids; done | tee gpurun_out/r2/gn_bench.txt
