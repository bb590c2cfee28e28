set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v3.log 2>&1; echo "ops rc=$?"
tail -3 gpurun_out/ops_test_v3.log
timeout 300 python tools/bench_ops.py 16 > gpurun_out/bench_ops_v3.txt 2>&1
timeout 300 python tools/unet_shapes.py 8 > gpurun_out/unet_shapes_v3.txt 2> gpurun_out/unet_shapes_v3.err
timeout 900 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/model_test_v3.log 2>&1; echo "model rc=$?"
tail -3 gpurun_out/model_test_v3.log
timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline > gpurun_out/bench_v3_short.json 2> gpurun_out/bench_v3_short.err
cat gpurun_out/bench_v3_short.json
