#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "audioldm_48k" > /tmp/t17.log 2>&1
grep -n "Error\|assert\|^E " /tmp/t17.log | cut -c1-300 | head -40 > gpurun_out/r2/run17_tests.log
cat gpurun_out/r2/run17_tests.log
