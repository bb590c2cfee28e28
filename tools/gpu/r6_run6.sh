#!/bin/bash
# round 6, call 6: f16x3 microbenchmark (tools/gpu/f16x3_probe.hip) + the fixed attention-through-LDS test
O=gpurun_out/r6_6; mkdir -p $O; export TMPDIR=/tmp
timeout 600 tools/gpu/f16x3_probe > $O/f16x3_accuracy.txt 2>&1; cat $O/f16x3_accuracy.txt
timeout 1200 python -m pytest tests/test_dma_gpu.py -q -m gpu -k "through_lds" -p no:cacheprovider 2>&1 | tail -5 > $O/tests_attn_lds.txt
cat $O/tests_attn_lds.txt
