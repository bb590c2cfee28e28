#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "attention or attn" 2>&1 | tail -3
for cfg in "1 2" "1 1" "0 2"; do set -- $cfg
  echo "== ALDM_ATTN_PIPE=$1 ALDM_ATTN_QT=$2"
  ALDM_ATTN_PIPE=$1 ALDM_ATTN_QT=$2 python tools/attn_ab.py 2>&1 | grep -v amdgpu.ids
done | tee gpurun_out/r2/attn_ab5.txt
