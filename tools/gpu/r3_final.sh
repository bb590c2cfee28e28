#!/bin/bash
# End-of-round run on the final sources: full GPU suite, smoke, PMC traffic of the igemm kernels (-> profiles/r03_pmc_traffic.json,
# stamped with the hash of the kernel sources), the default bench (which then reports that traffic), rocprofv3 kernel statistics
# and the per-grid trace of a 10-step job, the SQ counter table (two --pmc passes, no trace domains).
set -x
O=gpurun_out/r3/final
mkdir -p $O
rm -f gpurun_out/parity_report.txt
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/ -q -m gpu > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -4 $O/gpu_suite.log | cut -c1-200
cp gpurun_out/parity_report.txt $O/parity_report.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
QUICK="--steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe --no-strict --no-configs"
cd /tmp; export TMPDIR=/tmp
ALDM_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_fetch -- python $R/bench.py $QUICK > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_write -- python $R/bench.py $QUICK > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log
cp $O/pmc_traffic.json profiles/r03_pmc_traffic.json
timeout 900 python bench.py > $O/bench_final.json 2> $O/bench_final.err; tail -2 $O/bench_final.err; cut -c1-3000 $O/bench_final.json
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fin -o fin --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe --no-strict --no-configs > /dev/null 2>&1
cd $R
cp $(find /tmp/prof_fin -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
mkdir -p /tmp/kt && cp $(find /tmp/prof_fin -name "*kernel_trace.csv" | head -1) /tmp/kt/ && python tools/trace_by_grid.py /tmp/kt 90 > $O/trace_by_grid.txt 2>&1
python tools/trace_gaps.py /tmp/kt > $O/trace_gaps.txt 2>&1
head -12 $O/kernel_stats.csv | cut -c1-160
cd /tmp
KRE="igemm_dma|attention|layernorm|gn_partial|split_rows"
ALDM_NO_GRAPH=1 timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq1 -- python $R/bench.py $QUICK > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq2 -- python $R/bench.py $QUICK > /dev/null 2>&1
cd $R
python tools/pmc_sq_table.py /tmp/pmc_sq1 /tmp/pmc_sq2 > $O/pmc_sq.txt 2>&1; head -24 $O/pmc_sq.txt
