#!/bin/bash
# round 3, call 21: one-launch GroupNorm + split with the slab kept in registers between the statistics and the apply phase
# (ALDM_GN_KEEP=0: re-read, the previous form) — bitwise tests, per-shape times, same-box step A/B
mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests/test_dma_gpu.py tests/test_ops_gpu.py -q -m gpu -k "groupnorm or split_rows" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "unet or e2e_5step or vae" 2>&1 | tail -2
O=gpurun_out/r3/gn_keep_bench.txt; : > $O
for V in "ALDM_GN_KEEP=0" "ALDM_GN_KEEP=1"; do
  echo "## $V" >> $O
  env $V timeout 200 python tools/gn_bench.py --split 2>&1 | grep -v amdgpu.ids >> $O
done
grep -E "^##|weighted" $O
for i in 1 2; do
ALDM_GN_KEEP=0 timeout 300 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/GroupNorm slab re-read in the apply phase: /'
timeout 300 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/GroupNorm slab kept in registers:         /'
done | tee gpurun_out/r3/step_ab_gn_keep.txt
