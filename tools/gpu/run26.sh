set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v17.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/ops_test_v17.log
for S in 1 2; do
timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_gn$S.json 2> gpurun_out/bench_gn$S.err
python -c "import json;d=json.load(open('gpurun_out/bench_gn$S.json'));print('fused small GN run $S', d['value'], d['ms_per_step'], d['unet_step_ms'], d['unet_step_frac_of_f32_mfma_peak'])"
done
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "e2e_5step or unet or vae" > gpurun_out/model_test_gn.log 2>&1; echo "model rc=$?"; tail -2 gpurun_out/model_test_gn.log
