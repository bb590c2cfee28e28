#!/bin/bash
# round 6, call 31: the one-launch GroupNorm's size limit (elements per sample; default 2^20, chosen on isolated launches in round 2) scanned
# INSIDE the replayed step in both fp32-grade modes (call 30: 2^18 is 0.08 ms faster in bf16x6)
O=gpurun_out/r6_31; mkdir -p $O; export TMPDIR=/tmp
run() { env "$@" timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed "s/^/$*: /"; }
{
for MODE in bf16x6 f16x3; do
run ALDM_MMA=$MODE DEFAULTS=1
run ALDM_MMA=$MODE ALDM_GN_FUSED_MAX=65536
run ALDM_MMA=$MODE ALDM_GN_FUSED_MAX=131072
run ALDM_MMA=$MODE ALDM_GN_FUSED_MAX=262144
run ALDM_MMA=$MODE ALDM_GN_FUSED_MAX=393216
run ALDM_MMA=$MODE DEFAULTS=1
run ALDM_MMA=$MODE ALDM_GN_FUSED_MAX=262144
done
} > $O/step_ab_gn_fused_max.txt 2>&1; cat $O/step_ab_gn_fused_max.txt
