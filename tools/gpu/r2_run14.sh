#!/bin/bash
mkdir -p gpurun_out/r2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
for pipe in 1 0; do
ALDM_ATTN_PIPE=$pipe rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "attention" -f csv -d /tmp/pmc_attn_a$pipe -- python tools/attn_pmc.py > /tmp/pmc1.log 2>&1
ALDM_ATTN_PIPE=$pipe rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAVES SQ_INSTS_MFMA --kernel-include-regex "attention" -f csv -d /tmp/pmc_attn_b$pipe -- python tools/attn_pmc.py > /tmp/pmc2.log 2>&1
tail -n 3 /tmp/pmc2.log
echo "== pipe=$pipe"; python tools/pmc_summary.py /tmp/pmc_attn_a$pipe; python tools/pmc_summary.py /tmp/pmc_attn_b$pipe
done | tee gpurun_out/r2/attn_pmc2.txt
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > gpurun_out/r2/sq_counters.txt
