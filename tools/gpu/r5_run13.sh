#!/bin/bash
# round 5, call 13/14: single-row bisect of the exact-max kernels' non-finite outputs under extreme logits (tools/attn_extreme.py):
# chosen tile maxima, then the position map (which key positions does the running maximum miss?)
O=gpurun_out/r5_13; mkdir -p $O; export TMPDIR=/tmp
ALDM_ATTN_SCHED=1 timeout 300 python tools/attn_extreme.py 2>&1 | grep "maxima \[\|key tile\|Error\|error\|Traceback" > $O/attn_extreme_rows.txt; tail -12 $O/attn_extreme_rows.txt
