set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v11.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/ops_test_v11.log
timeout 1500 python tools/igemm_autotune.py gpurun_out/mi355x_igemm.json audioldm2-full audioldm_48k audioldm2-speech-gigaspeech > gpurun_out/autotune.log 2>&1; echo "tune rc=$?"; tail -2 gpurun_out/autotune.log
cp gpurun_out/mi355x_igemm.json audioldm2_amd/tuning/mi355x_igemm.json
timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_kg_tuned.json 2> gpurun_out/bench_kg_tuned.err
python -c "import json;d=json.load(open('gpurun_out/bench_kg_tuned.json'));print('kgroups + new table', d['value'], d['ms_per_step'], d['unet_step_ms'])"
timeout 1800 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/model_test_v11.log 2>&1; echo "model rc=$?"; tail -3 gpurun_out/model_test_v11.log
