#!/bin/bash
# round 6, call 32: rows per block of the operand-stationary GEMM (default: one block per CU and one round) re-checked INSIDE the replayed step
O=gpurun_out/r6_32; mkdir -p $O; export TMPDIR=/tmp
run() { env "$@" timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed "s/^/$*: /"; }
{
for MODE in bf16x6 f16x3; do
run ALDM_MMA=$MODE DEFAULTS=1
run ALDM_MMA=$MODE ALDM_OS_ROWS=64
run ALDM_MMA=$MODE ALDM_OS_ROWS=96
run ALDM_MMA=$MODE ALDM_OS_ROWS=256
run ALDM_MMA=$MODE DEFAULTS=1
done
} > $O/step_ab_os_rows.txt 2>&1; cat $O/step_ab_os_rows.txt
