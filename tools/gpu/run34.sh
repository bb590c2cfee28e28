mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/bx_ops.log 2>&1; echo "ops rc=$?"; tail -15 gpurun_out/bx_ops.log
timeout 600 python tools/mma_ab.py 10 2>&1 | tee gpurun_out/mma_ab.txt | cut -c1-260
