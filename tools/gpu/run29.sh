set -x
mkdir -p gpurun_out
for S in 1 2 1 2; do
ALDM_CFG_GRAPHS=$S timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_g$S.json 2> gpurun_out/bench_g$S.err
tail -2 gpurun_out/bench_g$S.err | grep -v amdgpu
python -c "import json;d=json.load(open('gpurun_out/bench_g$S.json'));print('cfg_graphs=$S', d['value'], d['ms_per_step'], d['unet_step_ms'], d['unet_step_frac_of_f32_mfma_peak'])"
done
ALDM_CFG_GRAPHS=2 timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "e2e_5step" 2>&1 | tail -3
