#!/bin/bash
# round 6, call 20: does the round-5 operand-stationary kernel beat the 2-part tables' (round-4 tuned) choices?  tools/os_probe.py in bf16x3 (the
# 2-part geometry tables also serve f16x3)
O=gpurun_out/r6_20; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/os_probe.py bf16x3 2>&1 | grep -v amdgpu.ids | tee $O/os_probe_bf16x3.txt
