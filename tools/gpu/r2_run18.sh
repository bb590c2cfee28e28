#!/bin/bash
mkdir -p gpurun_out/r2
{
python tools/dma_ablate_shapes.py
for v in nost k1nost; do ALDM_LIB_PATH=tools/gpu/libaldm_$v.so python tools/dma_ablate_shapes.py; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/dma_ablate_shapes3.txt
cat gpurun_out/r2/dma_ablate_shapes3.txt
