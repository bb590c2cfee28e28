#!/bin/bash
mkdir -p gpurun_out/r2
python -m pytest tests/test_ops_gpu.py tests/test_dma_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r2/run15_tests.log
python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/r2/bench15.err | tee gpurun_out/r2/bench15.json
