#!/bin/bash
# attention kernel after the VALU diet (buffer loads, VGPR-form MFMA, log2-domain softmax), fast GELU + hardware bf16
# conversion in the GEMM epilogues: unit tests, attention A/B, the short-K GEMM shapes again
mkdir -p gpurun_out/r2
python -m pytest tests/test_ops_gpu.py tests/test_dma_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r2/run11_tests.log
cat gpurun_out/r2/run11_tests.log
python tools/attn_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/attn_ab2.txt
python tools/dma_ablate_shapes.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/dma_shapes2.txt
