#!/bin/bash
# round 3, call 22: short GEMMs on cold caches (weights / activations not L2- or MALL-resident) vs the tuner's hot loops
mkdir -p gpurun_out/r3
timeout 600 python tools/cold_gemm_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/cold_gemm_probe.txt
