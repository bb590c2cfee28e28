#!/bin/bash
mkdir -p gpurun_out/r2
python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2> gpurun_out/r2/bench21.err | tee gpurun_out/r2/bench21.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['unet_step_ms'], d['roofline']['all_igemm_ms'], [ (k['kernel'],k['ms']) for k in d['roofline']['top_kernels']])"
