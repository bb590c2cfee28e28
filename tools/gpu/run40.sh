set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
timeout 300 python tools/eager_probe.py 2>&1 | grep "ms/pass"
timeout 900 python bench.py > gpurun_out/bench_final_bx.json 2> gpurun_out/bench_final_bx.err; tail -2 gpurun_out/bench_final_bx.err; cat gpurun_out/bench_final_bx.json | cut -c1-1500
timeout 600 python bench.py --mma f32 --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_final_f32.json 2>/dev/null; cat gpurun_out/bench_final_f32.json | cut -c1-600
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_fin -o fin --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
cp $(find /tmp/prof_fin -name "*kernel_stats.csv" | head -1) $R/gpurun_out/kernel_stats_bx.csv
cp $(find /tmp/prof_fin -name "*kernel_trace.csv" | head -1) $R/gpurun_out/kernel_trace_bx.csv
head -12 $R/gpurun_out/kernel_stats_bx.csv | cut -c1-180
ALDM_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex igemm_kernel -f csv -d $R/gpurun_out/pmc_fetch_bx -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex igemm_kernel -f csv -d $R/gpurun_out/pmc_write_bx -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py gpurun_out/pmc_fetch_bx gpurun_out/pmc_write_bx gpurun_out/pmc_traffic_bx.json > /dev/null 2>&1; ls -la gpurun_out/pmc_traffic_bx.json
find gpurun_out/pmc_fetch_bx gpurun_out/pmc_write_bx -name "*.csv" -size +1M -delete
