#!/bin/bash
# round 4, call 2: the operand-stationary DMA GEMM — bitwise tests against the classic kernel, then the graph-timed probe in both
# product modes
mkdir -p gpurun_out/r4/run2
timeout 900 python -m pytest tests/test_dma_gpu.py -q -m gpu -x -k "dma_os" 2>&1 | tail -15
for M in bf16x6 bf16x3; do
  timeout 600 python tools/os_probe.py $M --rows 2>&1 | tee gpurun_out/r4/run2/os_probe_$M.txt | cut -c1-400
done
