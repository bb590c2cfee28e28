#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" 2>&1 | tail -3
python tools/attn_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/attn_ab3.txt
