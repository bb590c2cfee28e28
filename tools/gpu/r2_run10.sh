#!/bin/bash
# ablation of the short-K transformer GEMMs (tools/dma_ablate_shapes.py) with the shipped library and five debug builds
mkdir -p gpurun_out/r2
{
python tools/dma_ablate_shapes.py
for v in epi16 k1 k1epi nomfma nodma; do ALDM_LIB_PATH=tools/gpu/libaldm_$v.so python tools/dma_ablate_shapes.py; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/dma_ablate_shapes.txt
cat gpurun_out/r2/dma_ablate_shapes.txt
