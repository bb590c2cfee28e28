set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v1.log 2>&1; echo "ops rc=$?" 
tail -5 gpurun_out/ops_test_v1.log
timeout 600 python tools/igemm_tune.py > gpurun_out/tune_v1.txt 2>&1
timeout 300 python tools/unet_shapes.py 8 > gpurun_out/unet_shapes_v1.txt 2> gpurun_out/unet_shapes_v1.err
timeout 900 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/model_test_v1.log 2>&1; echo "model rc=$?"
tail -5 gpurun_out/model_test_v1.log
