#!/bin/bash
# round 4, call 1: rocprofv3 evidence for the fp32-grade (bf16x6) product mode on the round-3 kernels:
# kernel statistics + per-grid trace of a 10-step job, SQ counter table, FETCH_SIZE / WRITE_SIZE traffic.  Counter passes carry no
# trace domains besides --kernel-trace.
set -x
O=gpurun_out/r4/run1
mkdir -p $O
R=$GRAFT_REPO_ROOT
MODE=${MODE:-bf16x6}
QUICK="--mma $MODE --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe --no-strict --no-configs"
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fin -o fin --output-format csv -- python $R/bench.py --mma $MODE --steps 1 --warmup 0 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe --no-strict --no-configs > /dev/null 2>&1
cd $R
cp $(find /tmp/prof_fin -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$MODE.csv
mkdir -p /tmp/kt && cp $(find /tmp/prof_fin -name "*kernel_trace.csv" | head -1) /tmp/kt/ && python tools/trace_by_grid.py /tmp/kt 100 > $O/trace_by_grid_$MODE.txt 2>&1
head -14 $O/kernel_stats_$MODE.csv | cut -c1-150
cd /tmp
KRE="igemm_dma|attention|layernorm|gn_partial|split_rows"
ALDM_NO_GRAPH=1 timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq1 -- python $R/bench.py $QUICK > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq2 -- python $R/bench.py $QUICK > /dev/null 2>&1
cd $R
python tools/pmc_sq_table.py /tmp/pmc_sq1 /tmp/pmc_sq2 > $O/pmc_sq_$MODE.txt 2>&1; head -30 $O/pmc_sq_$MODE.txt
cd /tmp
ALDM_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_fetch -- python $R/bench.py $QUICK > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_write -- python $R/bench.py $QUICK > /dev/null 2>&1
cd $R
python tools/pmc_traffic.py /tmp/pmc_fetch /tmp/pmc_write $O/pmc_traffic_$MODE.json > $O/pmc_traffic.log 2>&1; tail -3 $O/pmc_traffic.log
# the strict-mode roofline line of the same sources (dominant instantiation, event timed)
timeout 600 python bench.py --mma $MODE --steps 2 --warmup 1 --no-cpu-baseline --no-strict --no-configs > $O/bench_$MODE.json 2> $O/bench_$MODE.err; tail -2 $O/bench_$MODE.err; cut -c1-2500 $O/bench_$MODE.json
