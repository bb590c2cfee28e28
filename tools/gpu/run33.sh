mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/w8_ops2.log 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/w8_ops2.log
timeout 900 python tools/wave8_ab.py audioldm2-full 0 1 3 5 7 2>&1 | grep "wave8 mask" | tee gpurun_out/w8_masks.txt
