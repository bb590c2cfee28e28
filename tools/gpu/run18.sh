set -x
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/model_test_v14.log 2>&1; echo "model rc=$?"; tail -4 gpurun_out/model_test_v14.log
timeout 1500 python bench.py --no-cpu-baseline > gpurun_out/bench_v14.json 2> gpurun_out/bench_v14.err
python -c "import json;d=json.load(open('gpurun_out/bench_v14.json'));print('cached-graph bench', d['value'], d['ms_per_step'], d['unet_step_ms'], d['unet_step_frac_of_f32_mfma_peak'])"
ALDM_NO_GRAPH_CACHE=1 timeout 1500 python bench.py --no-cpu-baseline --no-roofline > gpurun_out/bench_v14_nocache.json 2> gpurun_out/bench_v14_nocache.err
python -c "import json;d=json.load(open('gpurun_out/bench_v14_nocache.json'));print('no-cache bench', d['value'], d['ms_per_step'], d['unet_step_ms'])"
