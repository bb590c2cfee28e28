#!/bin/bash
# round 6, call 18: f16x3 against bf16x6 on a UNet whose every parameter tensor has its own gain (2^-2.5 .. 2^2.5); the saturation test
O=gpurun_out/r6_18; mkdir -p $O; export TMPDIR=/tmp
ALDM_ERR_LOG=$O/err_log.tsv timeout 900 python -m pytest tests/test_f16x3_gpu.py -q -m gpu -p no:cacheprovider -k "gains or saturate" 2>&1 | grep -v "^        \|^    def\|^$" | tail -30 | tee $O/tests.txt
cat $O/err_log.tsv 2>/dev/null | cut -c1-200
