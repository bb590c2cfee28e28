#!/bin/bash
# round 3, call 19: attention loop with two score / K register sets (no end-of-tile copies) — bitwise tests, same-box step A/B against
# a build with the copies (tools/gpu/libaldm_nopp.so); one-launch GroupNorm with 512 blocks, per shape and in the step
mkdir -p gpurun_out/r3
timeout 600 python -m pytest tests/test_dma_gpu.py tests/test_ops_gpu.py -q -m gpu -k "qkv_epilogue or groupnorm or attention" 2>&1 | tail -2
O=gpurun_out/r3/gn_split_bench2.txt; : > $O
for V in "ALDM_GN_MIN_BLOCKS=512" "ALDM_GN_MIN_BLOCKS=512 ALDM_GN_SPLIT_FUSED=0"; do
  echo "## $V" >> $O
  env $V timeout 200 python tools/gn_bench.py --split 2>&1 | grep -v amdgpu.ids >> $O
done
grep -E "^##|weighted" $O
for i in 1 2; do
ALDM_LIB_PATH=tools/gpu/libaldm_nopp.so timeout 300 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/attention: registers copied per tile:  /'
timeout 300 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/attention: two register sets:          /'
ALDM_GN_MIN_BLOCKS=512 timeout 300 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/two sets + GroupNorm on 512 blocks:     /'
done | tee gpurun_out/r3/step_ab_attn_pingpong.txt
