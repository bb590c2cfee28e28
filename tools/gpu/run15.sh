set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v12.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/ops_test_v12.log
timeout 300 python tools/bench_ops.py 16 2>/dev/null | grep "layernorm\|gn_stats"
for S in 0 1; do
ALDM_CFG_STREAMS=$S timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_s$S.json 2> gpurun_out/bench_s$S.err
python -c "import json;d=json.load(open('gpurun_out/bench_s$S.json'));print('streams=$S', d['value'], d['ms_per_step'], d['unet_step_ms'], d['unet_step_frac_of_f32_mfma_peak'])"
done
timeout 1800 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/model_test_v12.log 2>&1; echo "model rc=$?"; tail -3 gpurun_out/model_test_v12.log
