#!/bin/bash
# round 3, call 24: ONE weight prefetcher beside the pass, paced by the GEMM kernels' progress marks — model parity through the
# captured step, then same-box step A/B by lead (0 = off) and prefetcher size
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "e2e_5step or cached_step_graph or timesteps_subset" 2>&1 | tail -3
for i in 1 2; do
for V in "ALDM_PREFETCH_LEAD=0" "ALDM_PREFETCH_LEAD=1" "ALDM_PREFETCH_LEAD=2" "ALDM_PREFETCH_LEAD=4" "ALDM_PREFETCH_LEAD=2 ALDM_PREFETCH_BLOCKS=32" "ALDM_PREFETCH_LEAD=2 ALDM_PREFETCH_BLOCKS=8"; do
env $V timeout 300 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed "s/^/$V: /"
done
done | tee gpurun_out/r3/step_ab_prefetch2.txt
