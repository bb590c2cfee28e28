#!/bin/bash
# Decode kernels per shape (graph timed, HBM-cold weights): shipped 32-column blocks vs a 16-column build.
set -x
O=gpurun_out/r4/run9
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 5 120 python tools/decode_shapes.py < /dev/null 2>&1 | grep -v amdgpu.ids | tee $O/decode_shapes_ct32.txt
ALDM_LIB_PATH=tools/gpu/libaldm_ct16.so timeout -k 5 120 python tools/decode_shapes.py < /dev/null 2>&1 | grep -v amdgpu.ids | tee $O/decode_shapes_ct16.txt
