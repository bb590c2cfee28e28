#!/bin/bash
# round 5, call 11: WHAT bounds the pre-split attention loop?  Call 10 took 26 % of its VALU instructions out (one-pass fixed-reference
# kernel) for -3.5 % of its time.  Timing-only ablation builds of that kernel (ALDM_ATTN3_ABLATE: 1 every K / V load reads tile 0,
# 2 no operand splits, 4 no MFMAs, 10 MFMAs + loads only, 11 MFMAs only with cache-resident loads); the round-5 build variants of the
# exact-max kernel re-measured WITH attn.hip's per-file flag (build_variant.sh dropped -amdgpu-mfma-vgpr-form=1 until now);
# extreme-logit diagnostic (tools/attn_extreme.py) for the three pre-split kernels and the fp32-K/V path
O=gpurun_out/r5_11; mkdir -p $O; export TMPDIR=/tmp
{
timeout 300 python tools/attn_probe.py 2>&1 | grep " us " | sed 's/^/full kernel: /'
for A in 1 2 4 10 11; do
ALDM_LIB_PATH=tools/gpu/libaldm_attn3_abl$A.so timeout 300 python tools/attn_probe.py 2>&1 | grep " us " | sed "s/^/ABLATE=$A: /"
done
ALDM_ATTN_SCHED=1 timeout 300 python tools/attn_probe.py 2>&1 | grep " us " | sed 's/^/exact-max kernel (sched=1): /'
for V in noslp mvq0 mvq4; do
ALDM_ATTN_SCHED=1 ALDM_LIB_PATH=tools/gpu/libaldm_attn_$V.so timeout 300 python tools/attn_probe.py 2>&1 | grep " us " | sed "s/^/exact-max kernel, variant $V: /"
done
} > $O/attn_ablate.txt 2>&1; cat $O/attn_ablate.txt
{
timeout 300 python tools/attn_extreme.py 2>&1 | grep "jump"
ALDM_ATTN_SCHED=1 timeout 300 python tools/attn_extreme.py 2>&1 | grep "jump"
ALDM_ATTN_SCHED=0 timeout 300 python tools/attn_extreme.py 2>&1 | grep "jump"
} > $O/attn_extreme.txt 2>&1; cat $O/attn_extreme.txt
