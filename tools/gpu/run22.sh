set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_final.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/ops_test_final.log
timeout 1800 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/model_test_final.log 2>&1; echo "model rc=$?"; tail -2 gpurun_out/model_test_final.log
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
ALDM_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex igemm_kernel -f csv -d $R/gpurun_out/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > $R/gpurun_out/pmc_fetch.log 2>&1
ALDM_NO_GRAPH=1 timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex igemm_kernel -f csv -d $R/gpurun_out/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > $R/gpurun_out/pmc_write.log 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r01_pmc_traffic.json > gpurun_out/pmc_traffic.log 2>&1
cp profiles/r01_pmc_traffic.json gpurun_out/r01_pmc_traffic.json
find gpurun_out/pmc_fetch gpurun_out/pmc_write -name "*.csv" -size +1M -delete
timeout 1500 python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
cat gpurun_out/bench_final.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_final -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline > $R/gpurun_out/prof_final.log 2>&1
cd $R
find /tmp/prof_final -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_final_kernel_stats.csv
python tools/trace_by_grid.py /tmp/prof_final 70 > gpurun_out/trace_by_grid_final.txt 2>&1
