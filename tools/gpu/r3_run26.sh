#!/bin/bash
# round 3, call 26: the per-node floor of a HIP-graph replay (chains of near-empty dependent launches)
mkdir -p gpurun_out/r3
timeout 300 python tools/graph_floor_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/graph_floor_probe.txt
