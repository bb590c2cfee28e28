set -x
mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_dma_gpu.py -q -m gpu -x > gpurun_out/r2/dma_test5.log 2>&1; echo "dma tests rc=$?"; tail -6 gpurun_out/r2/dma_test5.log
timeout 200 python tools/x3_accuracy.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/x3_accuracy.txt
timeout 300 python tools/dma_ablate.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/dma_tapinner.txt
ALDM_MMA=bf16x3 timeout 300 python tools/dma_ablate.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2/dma_tapinner.txt
timeout 900 python tools/ab_libs.py --tests none --reps 2 default default::ALDM_MMA=bf16x3 2>&1 | tee gpurun_out/r2/step_ab3.txt
ALDM_MMA=bf16x3 timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "unet_full or e2e_full_5step or e2e_200step" -s 2>&1 | grep -i "rms\|passed\|failed\|error" | tee gpurun_out/r2/model_x3.log
