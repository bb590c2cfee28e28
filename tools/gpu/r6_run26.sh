#!/bin/bash
# round 6, call 26: the 3-part (bf16x6) geometry table tuned INSIDE the replayed step (tools/instep_autotune.py): one geometry's entry at a time,
# kept only when the step itself gets faster twice in a row
O=gpurun_out/r6_26; mkdir -p $O; export TMPDIR=/tmp
INSTEP_BUDGET_S=1500 ALDM_MMA=bf16x6 timeout 2400 python tools/instep_autotune.py $O/instep_bf16x6.json audioldm2-full 40 2>&1 | grep -v amdgpu.ids | tee $O/instep_autotune_bf16x6.txt | tail -50
