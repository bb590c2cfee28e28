set -x
mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_dma_gpu.py -q -m gpu > gpurun_out/r2/dma_test2.log 2>&1; echo "dma tests rc=$?"; tail -5 gpurun_out/r2/dma_test2.log
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k groupnorm > gpurun_out/r2/gn_test.log 2>&1; echo "gn tests rc=$?"; tail -5 gpurun_out/r2/gn_test.log
timeout 120 python tools/mfma_peak.py 20000 2>&1 | tee gpurun_out/r2/mfma_peak.txt
for m in default 1 2 3 4 8 12; do
  if [ $m = default ]; then timeout 120 python tools/dma_ablate.py; else ALDM_LIB_PATH=tools/gpu/libaldm_dmaabl$m.so timeout 120 python tools/dma_ablate.py; fi
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r2/dma_ablate.txt
timeout 400 python tools/dma_bench.py 10 > gpurun_out/r2/dma_bench2.txt 2>&1; echo "bench rc=$?"; grep -o "^.\{28\}\|old.\{30\}\|best.*" gpurun_out/r2/dma_bench2.txt | paste - - - 
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "unet_full or e2e_full_5step or cfg_batched or cached" > gpurun_out/r2/model_dma.log 2>&1; echo "model(dma) rc=$?"; tail -5 gpurun_out/r2/model_dma.log
timeout 900 python tools/ab_libs.py --tests none --reps 2 default::ALDM_DMA=0 default::ALDM_DMA=1 default::ALDM_DMA=1,ALDM_ATTN_MMA=bf16x6 default::ALDM_DMA=0,ALDM_ATTN_MMA=bf16x6 2>&1 | tee gpurun_out/r2/step_ab.txt
