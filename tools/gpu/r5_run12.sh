#!/bin/bash
# round 5, call 12: the two ablations call 11 could not load (MFMAs + loads only: 10; MFMAs only, cache-resident loads: 11), four
# accumulator chains instead of two (27 = MFMAs only; 16 = the full loop, wrong results); single-row bisect of the exact-max kernels'
# non-finite outputs under extreme logits (tools/attn_extreme.py)
O=gpurun_out/r5_12; mkdir -p $O; export TMPDIR=/tmp
{
timeout 300 python tools/attn_probe.py 2>&1 | grep " us " | sed 's/^/full kernel: /'
for A in 10 11 27 16; do
ALDM_LIB_PATH=tools/gpu/libaldm_attn3_abl$A.so timeout 300 python tools/attn_probe.py 2>&1 | grep " us \|Error\|error" | sed "s/^/ABLATE=$A: /"
done
} > $O/attn_ablate2.txt 2>&1; cat $O/attn_ablate2.txt
ALDM_ATTN_SCHED=1 timeout 300 python tools/attn_extreme.py 2>&1 | grep "maxima" > $O/attn_extreme_rows.txt; cat $O/attn_extreme_rows.txt
