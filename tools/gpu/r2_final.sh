#!/bin/bash
# End-of-round run: full GPU suite, smoke, PMC traffic of the igemm kernels (-> profiles/r02_pmc_traffic.json, stamped with the
# hash of the kernel sources), the default bench (which then reports that traffic), rocprofv3 kernel statistics of a 10-step job.
set -x
mkdir -p gpurun_out/r2
rm -f gpurun_out/parity_report.txt
R=$GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r2/gpu_suite_final.log 2>&1; echo "gpu suite rc=$?"; tail -4 gpurun_out/r2/gpu_suite_final.log | cut -c1-200
cp gpurun_out/parity_report.txt gpurun_out/r2/parity_report_final.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
cd /tmp; export TMPDIR=/tmp
ALDM_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_fetch -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_write -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write gpurun_out/r2/pmc_traffic_final.json > gpurun_out/r2/pmc_traffic_final.log 2>&1; tail -3 gpurun_out/r2/pmc_traffic_final.log
cp gpurun_out/r2/pmc_traffic_final.json profiles/r02_pmc_traffic.json
timeout 1200 python bench.py > gpurun_out/r2/bench_final.json 2> gpurun_out/r2/bench_final.err; tail -2 gpurun_out/r2/bench_final.err; cut -c1-2500 gpurun_out/r2/bench_final.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_fin -o fin --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
cd $R
cp $(find /tmp/prof_fin -name "*kernel_stats.csv" | head -1) gpurun_out/r2/kernel_stats_final.csv
mkdir -p /tmp/kt && cp $(find /tmp/prof_fin -name "*kernel_trace.csv" | head -1) /tmp/kt/ && python tools/trace_by_grid.py /tmp/kt 90 > gpurun_out/r2/trace_by_grid_final.txt 2>&1
head -12 gpurun_out/r2/kernel_stats_final.csv | cut -c1-160
