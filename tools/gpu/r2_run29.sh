#!/bin/bash
mkdir -p gpurun_out/r2
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
ALDM_NO_GRAPH=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "igemm_dma|attention|layernorm|gn_partial|split_rows" -f csv -d /tmp/pmc_sq1 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-include-regex "igemm_dma|attention|layernorm|gn_partial|split_rows" -f csv -d /tmp/pmc_sq2 -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
cd $R
{ echo "# pass 1"; python tools/pmc_summary.py /tmp/pmc_sq1; echo "# pass 2"; python tools/pmc_summary.py /tmp/pmc_sq2; } > gpurun_out/r2/pmc_sq_final.txt
wc -l gpurun_out/r2/pmc_sq_final.txt; head -60 gpurun_out/r2/pmc_sq_final.txt
