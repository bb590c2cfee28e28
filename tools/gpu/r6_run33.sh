#!/bin/bash
# round 6, call 33: the shipped GroupNorm size limit (2^17) against the old one (2^20, via the env override) on the final tree, alternating
O=gpurun_out/r6_33; mkdir -p $O; export TMPDIR=/tmp
run() { env "$@" timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed "s/^/$*: /"; }
{
for i in 1 2; do
run ALDM_MMA=bf16x6 ALDM_GN_FUSED_MAX=1048576
run ALDM_MMA=bf16x6 SHIPPED=1
done
run ALDM_MMA=f16x3 ALDM_GN_FUSED_MAX=1048576
run ALDM_MMA=f16x3 SHIPPED=1
} > $O/step_ab_gn_limit_shipped.txt 2>&1; cat $O/step_ab_gn_limit_shipped.txt
