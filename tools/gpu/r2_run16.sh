#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests -x -q -m gpu > /tmp/t16.log 2>&1
grep -n "Fatal\|fault\|FAILED\|passed\|failed" /tmp/t16.log | cut -c1-250 | head -30 > gpurun_out/r2/run16_tests.log
tail -n 15 /tmp/t16.log | cut -c1-400 >> gpurun_out/r2/run16_tests.log
cat gpurun_out/r2/run16_tests.log
