set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v7.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/ops_test_v7.log
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/model_test_v7.log 2>&1; echo "model rc=$?"
tail -3 gpurun_out/model_test_v7.log
timeout 900 python bench.py --steps 1 --warmup 0 --ddim-steps 20 --no-cpu-baseline > gpurun_out/bench_v7_short.json 2> gpurun_out/bench_v7_short.err
cat gpurun_out/bench_v7_short.json
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_v7 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline > $R/gpurun_out/prof_v7.log 2>&1
cd $R
find gpurun_out/prof_v7 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_v7_kernel_stats.csv
find gpurun_out/prof_v7 -name "*kernel_trace.csv" -delete
head -20 gpurun_out/prof_v7_kernel_stats.csv | cut -c1-160
