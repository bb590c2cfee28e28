set -x
mkdir -p gpurun_out
ALDM_IGEMM_STAGES=2 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v16.log 2>&1; echo "ops(stages2) rc=$?"; tail -2 gpurun_out/ops_test_v16.log
for S in 1 2 1 2; do
ALDM_IGEMM_STAGES=$S timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline > gpurun_out/bench_st$S.json 2> gpurun_out/bench_st$S.err
python -c "import json;d=json.load(open('gpurun_out/bench_st$S.json'));print('stages=$S', d['value'], d['ms_per_step'], d['unet_step_ms'], d['unet_step_frac_of_f32_mfma_peak'], d['roofline']['all_igemm_tflops'])"
done
for S in 1 2; do ALDM_IGEMM_STAGES=$S timeout 300 python tools/bench_ops.py 16 2>/dev/null | grep "conv3x3\|linear" | sed "s/^/st$S /" ; done
