set -x
mkdir -p gpurun_out/r2
rm -f gpurun_out/parity_report.txt
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r2/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -6 gpurun_out/r2/gpu_suite.log
cp gpurun_out/parity_report.txt gpurun_out/r2/parity_report.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err; tail -2 gpurun_out/r2/bench.err; cut -c1-3000 gpurun_out/r2/bench.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_fin -o fin --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
cp $(find /tmp/prof_fin -name "*kernel_stats.csv" | head -1) $R/gpurun_out/r2/kernel_stats.csv
cp $(find /tmp/prof_fin -name "*kernel_trace.csv" | head -1) $R/gpurun_out/r2/kernel_trace.csv
head -30 $R/gpurun_out/r2/kernel_stats.csv | cut -c1-200
ALDM_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write gpurun_out/r2/pmc_traffic.json > /dev/null 2>&1; ls -la gpurun_out/r2/pmc_traffic.json
mkdir -p /tmp/kt && cp gpurun_out/r2/kernel_trace.csv /tmp/kt/ && python tools/trace_by_grid.py /tmp/kt 70 > gpurun_out/r2/trace_by_grid.txt 2>&1; head -50 gpurun_out/r2/trace_by_grid.txt
rm -f gpurun_out/r2/kernel_trace.csv
