set -x
mkdir -p gpurun_out/r2
timeout 600 python -m pytest tests/test_dma_gpu.py -x -q -m gpu > gpurun_out/r2/dma_test.log 2>&1; echo "dma tests rc=$?"; tail -15 gpurun_out/r2/dma_test.log
timeout 500 python tools/dma_bench.py 20 > gpurun_out/r2/dma_bench.txt 2>&1; echo "bench rc=$?"; cut -c1-330 gpurun_out/r2/dma_bench.txt
timeout 200 python tools/attn_ab.py 2>&1 | tee gpurun_out/r2/attn_ab.txt
ALDM_EXPERIMENTAL=1 timeout 600 python -m pytest tests/test_seqgen_gpu.py -q -m gpu > gpurun_out/r2/seqgen_test.log 2>&1; echo "seqgen rc=$?"; tail -30 gpurun_out/r2/seqgen_test.log
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/r2/ops_test.log 2>&1; echo "ops rc=$?"; tail -5 gpurun_out/r2/ops_test.log
