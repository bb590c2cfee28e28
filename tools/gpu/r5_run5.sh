#!/bin/bash
# round 5, call 5: where the step's time goes now — rocprofv3 kernel trace of a 10-step job (per-grid table) + the SQ counter table
O=gpurun_out/r5_5; mkdir -p $O; R=$GRAFT_REPO_ROOT
Q="--steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-configs --no-conditioners --no-api-default"
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fin -o fin --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-configs --no-conditioners --no-api-default > /dev/null 2>&1
cd $R
cp $(find /tmp/prof_fin -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16x6.csv
mkdir -p /tmp/kt && cp $(find /tmp/prof_fin -name "*kernel_trace.csv" | head -1) /tmp/kt/ && python tools/trace_by_grid.py /tmp/kt 100 > $O/trace_by_grid_bf16x6.txt 2>&1
head -60 $O/trace_by_grid_bf16x6.txt
cd /tmp
KRE="igemm_dma|attention|layernorm|gn_partial|split_rows"
ALDM_NO_GRAPH=1 timeout 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq1 -- python $R/bench.py $Q > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq2 -- python $R/bench.py $Q > /dev/null 2>&1
cd $R
python tools/pmc_sq_table.py /tmp/pmc_sq1 /tmp/pmc_sq2 > $O/pmc_sq_bf16x6.txt 2>&1; head -30 $O/pmc_sq_bf16x6.txt
