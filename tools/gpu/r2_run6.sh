set -x
mkdir -p gpurun_out/r2
timeout 300 python -m pytest tests/test_phoneme.py tests/test_dma_gpu.py -q -m gpu 2>&1 | tail -8
timeout 900 python tools/ab_libs.py --tests none --reps 2 default tools/gpu/libaldm_tapinner.so 2>&1 | tee gpurun_out/r2/dma_taporder_ab.txt
ALDM_MMA=bf16x3 timeout 900 python tools/dma_autotune.py gpurun_out/r2/mi355x_igemm_dma_bf16x3.json audioldm2-full > gpurun_out/r2/dma_autotune_x3.txt 2>&1; echo "autotune rc=$?"; tail -3 gpurun_out/r2/dma_autotune_x3.txt
cp gpurun_out/r2/mi355x_igemm_dma_bf16x3.json audioldm2_amd/tuning/mi355x_igemm_dma_bf16x3.json
timeout 900 python tools/ab_libs.py --tests none --reps 2 default default::ALDM_MMA=bf16x3 2>&1 | tee gpurun_out/r2/step_ab4.txt
