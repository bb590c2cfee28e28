#!/bin/bash
# round 3, call 23: weight prefetch on a graph branch (ops.WeightPrefetch + aldm_prefetch): model parity through the captured step,
# then same-box step A/B by lead distance (0 = off)
mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -k "e2e_5step or cached_step_graph or timesteps_subset" 2>&1 | tail -3
for i in 1 2; do
for D in 0 2 4 8; do
ALDM_PREFETCH_DIST=$D timeout 300 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed "s/^/prefetch lead $D: /"
done
done | tee gpurun_out/r3/step_ab_prefetch.txt
