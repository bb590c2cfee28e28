#!/bin/bash
# round 6, call 14: split-K decode step of the sequence generator (7 whole-chip launches per GPT-2 block): op / generator tests,
# 512-token probe against the column-tile kernels and the general path
O=gpurun_out/r6_14; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_seqgen_gpu.py -q -m gpu -p no:cacheprovider 2>&1 | grep -v "^        \|^    def\|^$" | tail -30 > $O/tests_seqgen.txt
tail -12 $O/tests_seqgen.txt
timeout 900 python tools/decode_probe.py 512 2>&1 | grep -v amdgpu.ids > $O/decode_probe.txt; cat $O/decode_probe.txt
