#!/bin/bash
O=gpurun_out/r6_8; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_f16x3_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^        \|^    def\|^$" | tail -60 > $O/tests_f16x3.txt
cat $O/tests_f16x3.txt
