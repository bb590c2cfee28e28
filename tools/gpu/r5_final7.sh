#!/bin/bash
# round 5, final call 7 (after the row-maximum fix, final kernel sources): per-launch probe of the three pre-split attention loops in both
# product modes on one box; the extreme-logit diagnostic again (random rows + chosen tile maxima) for the default and the one-pass loop
O=gpurun_out/r5_final7; mkdir -p $O; export TMPDIR=/tmp
{
for S in 1 0 2; do ALDM_ATTN_SCHED=$S timeout 200 python tools/attn_probe.py 2>&1 | grep " us "; done
for S in 1 2; do ALDM_MMA=bf16x3 ALDM_ATTN_SCHED=$S timeout 200 python tools/attn_probe.py 2>&1 | grep " us "; done
} > $O/attn_probe_after_fix.txt 2>&1; cat $O/attn_probe_after_fix.txt | cut -c1-200
{
timeout 200 python tools/attn_extreme.py 2>&1 | grep "jump\|maxima \["
ALDM_ATTN_SCHED=2 timeout 200 python tools/attn_extreme.py 2>&1 | grep "jump\|maxima \["
} > $O/attn_extreme_after_fix.txt 2>&1; grep -c "non-finite pre-split 0 fp32-K/V 0\|fp32-K/V non-finite 0 .* pre-split non-finite 0" $O/attn_extreme_after_fix.txt; grep -vc "non-finite pre-split 0 fp32-K/V 0\|fp32-K/V non-finite 0 .* pre-split non-finite 0" $O/attn_extreme_after_fix.txt
