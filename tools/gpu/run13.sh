set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v10.log 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/ops_test_v10.log
ALDM_NO_TUNING=1 timeout 900 python tools/igemm_tune.py > gpurun_out/tune_v3.txt 2>&1
grep -v "^CSV" gpurun_out/tune_v3.txt | cut -c1-230
ALDM_NO_TUNING=1 timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline --no-roofline > gpurun_out/bench_kg.json 2> gpurun_out/bench_kg.err
python -c "import json;d=json.load(open('gpurun_out/bench_kg.json'));print('kgroups(cost model, no table)', d['value'], d['ms_per_step'], d['unet_step_ms'])"
