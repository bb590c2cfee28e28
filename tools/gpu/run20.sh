set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v15.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/ops_test_v15.log
ALDM_LIB_PATH=$GRAFT_REPO_ROOT/tools/gpu/libaldm_trace.so timeout 600 python tools/igemm_trace.py 2>&1 | grep "avg(mid)\|^---\|prologue" > gpurun_out/igemm_trace2.txt
cat gpurun_out/igemm_trace2.txt
timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline > gpurun_out/bench_ptr.json 2> gpurun_out/bench_ptr.err
python -c "import json;d=json.load(open('gpurun_out/bench_ptr.json'));print('running-pointer addr gen', d['value'], d['ms_per_step'], d['unet_step_ms'], d['unet_step_frac_of_f32_mfma_peak'], d['roofline']['all_igemm_tflops'])"
timeout 300 python tools/bench_ops.py 16 2>/dev/null | grep -v "attn\|device"
