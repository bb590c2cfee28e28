set -x
mkdir -p gpurun_out
for S in 0 1; do
ALDM_CFG_STREAMS=$S timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_streams$S.json 2> gpurun_out/bench_streams$S.err
tail -2 gpurun_out/bench_streams$S.err
python -c "import json;d=json.load(open('gpurun_out/bench_streams$S.json'));print('streams=$S', d['value'], d['ms_per_step'], d['unet_step_ms'])"
done
ALDM_CFG_STREAMS=1 timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "e2e_5step or cfg_batched or batch8" > gpurun_out/model_test_streams.log 2>&1; echo "model rc=$?"; tail -3 gpurun_out/model_test_streams.log
