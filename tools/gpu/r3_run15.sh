#!/bin/bash
mkdir -p gpurun_out/r3
timeout 600 python tools/branch_concurrency.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r3/branch_concurrency.txt
