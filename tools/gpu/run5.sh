set -x
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/model_test_v4.log 2>&1; echo "model rc=$?"
tail -3 gpurun_out/model_test_v4.log
timeout 1500 python bench.py > gpurun_out/bench_v4.json 2> gpurun_out/bench_v4.err
cat gpurun_out/bench_v4.json
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/prof_v4 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline > $R/gpurun_out/prof_v4.log 2>&1
cd $R
ls -R gpurun_out/prof_v4 | head
find gpurun_out/prof_v4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_v4_kernel_stats.csv
find gpurun_out/prof_v4 -name "*kernel_trace.csv" -delete
