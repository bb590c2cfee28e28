set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "ddpm or geglu or splitk" > gpurun_out/ops_test_v6.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/ops_test_v6.log
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/model_test_v6.log 2>&1; echo "model rc=$?"
tail -15 gpurun_out/model_test_v6.log
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
ALDM_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex igemm_kernel -f csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > $R/gpurun_out/pmc_fetch.log 2>&1
ALDM_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex igemm_kernel -f csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_traffic.json > gpurun_out/pmc_traffic.log 2>&1
tail -30 gpurun_out/pmc_traffic.log
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +1M -delete
timeout 900 python bench.py --model audioldm_48k --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline > gpurun_out/bench_48k_short.json 2> gpurun_out/bench_48k_short.err; tail -3 gpurun_out/bench_48k_short.err; cat gpurun_out/bench_48k_short.json
