set -x
mkdir -p gpurun_out
ALDM_DEEP_STREAMS=1 timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "e2e_5step or cfg_batched or batch8 or cached_step or unet_full" > gpurun_out/model_test_deep.log 2>&1; echo "model rc=$?"; tail -3 gpurun_out/model_test_deep.log
for S in 0 1 0 1; do
ALDM_DEEP_STREAMS=$S timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_deep$S.json 2> gpurun_out/bench_deep$S.err
python -c "import json;d=json.load(open('gpurun_out/bench_deep$S.json'));print('deep_streams=$S', d['value'], d['ms_per_step'], d['unet_step_ms'], d['unet_step_frac_of_f32_mfma_peak'])"
done
