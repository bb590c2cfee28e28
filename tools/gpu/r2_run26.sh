#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -2 | cut -c1-200
python tools/attn_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/attn_ab6.txt
