set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v13.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/ops_test_v13.log
timeout 600 python tools/igemm_tune.py 2>/dev/null | grep "pre=1" | cut -c1-120
timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_uni.json 2> gpurun_out/bench_uni.err
python -c "import json;d=json.load(open('gpurun_out/bench_uni.json'));print('UNI', d['value'], d['ms_per_step'], d['unet_step_ms'], d['unet_step_frac_of_f32_mfma_peak'])"
timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_uni2.json 2> gpurun_out/bench_uni2.err
python -c "import json;d=json.load(open('gpurun_out/bench_uni2.json'));print('UNI again', d['value'], d['ms_per_step'], d['unet_step_ms'], d['unet_step_frac_of_f32_mfma_peak'])"
