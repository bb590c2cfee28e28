set -x
mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_dma_gpu.py -q -m gpu > gpurun_out/r2/dma_test3.log 2>&1; echo "dma tests rc=$?"; tail -3 gpurun_out/r2/dma_test3.log
timeout 900 python tools/dma_autotune.py gpurun_out/r2/mi355x_igemm_dma.json audioldm2-full > gpurun_out/r2/dma_autotune.txt 2>&1; echo "autotune rc=$?"; tail -4 gpurun_out/r2/dma_autotune.txt
cp gpurun_out/r2/mi355x_igemm_dma.json audioldm2_amd/tuning/mi355x_igemm_dma.json
timeout 900 python tools/ab_libs.py --tests none --reps 2 default::ALDM_DMA=0,ALDM_ATTN_MMA=bf16x6 default::ALDM_DMA=1,ALDM_ATTN_MMA=bf16x6 2>&1 | tee gpurun_out/r2/step_ab2.txt
timeout 600 python -m pytest tests/test_reference_binding.py -q -m gpu -s 2>&1 | tail -4
