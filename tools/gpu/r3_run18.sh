#!/bin/bash
# round 3, call 18: the one-launch GroupNorm + split by block size / block count, against the two- and three-launch forms, per UNet shape
mkdir -p gpurun_out/r3
O=gpurun_out/r3/gn_split_bench.txt
: > $O
timeout 300 python -m pytest tests/test_dma_gpu.py tests/test_ops_gpu.py -q -m gpu -k "groupnorm or gn_" 2>&1 | tail -2
for V in "" "ALDM_GN_THREADS=512" "ALDM_GN_THREADS=1024" "ALDM_GN_THREADS=1024 ALDM_GN_MIN_BLOCKS=128" "ALDM_GN_THREADS=512 ALDM_GN_MIN_BLOCKS=128" "ALDM_GN_SPLIT_FUSED=0" "ALDM_GN_FUSED_MAX=0"; do
  echo "## ${V:-default (256 threads, 256 blocks)}" >> $O
  env $V timeout 200 python tools/gn_bench.py --split 2>&1 | grep -v amdgpu.ids >> $O
done
grep -E "^##|weighted" $O
