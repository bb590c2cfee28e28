#!/bin/bash
# Single-position decode kernels (csrc/decode.hip): op + generator tests, ABI load, 512-token generation A/B fast vs general.
set -x
O=gpurun_out/r4/run7
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout -k 5 300 python -m pytest tests/test_seqgen_gpu.py tests/test_abi.py -q -m gpu -x < /dev/null 2>&1 | tail -8 | tee $O/tests.log
timeout -k 5 240 python tools/decode_probe.py < /dev/null 2>&1 | grep -v amdgpu.ids | tee $O/decode_probe.txt
