#!/bin/bash
# round 3, call 25: the paced weight prefetcher with more lead / more workgroups (call 24: 1-4 launches of lead were late)
mkdir -p gpurun_out/r3
for V in "ALDM_PREFETCH_LEAD=0" "ALDM_PREFETCH_LEAD=8 ALDM_PREFETCH_BLOCKS=32" "ALDM_PREFETCH_LEAD=16 ALDM_PREFETCH_BLOCKS=32" "ALDM_PREFETCH_LEAD=32 ALDM_PREFETCH_BLOCKS=32" "ALDM_PREFETCH_LEAD=8 ALDM_PREFETCH_BLOCKS=64" "ALDM_PREFETCH_LEAD=16 ALDM_PREFETCH_BLOCKS=64" "ALDM_PREFETCH_LEAD=0"; do
env $V timeout 300 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed "s/^/$V: /"
done | tee gpurun_out/r3/step_ab_prefetch3.txt
