#!/bin/bash
# round 6, call 5: pre-split self-attention with K / V^T through LDS once per block (aldm_attention_sched 3) — bitwise test against the
# default kernel, per-launch probe, step A/B
O=gpurun_out/r6_5; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_dma_gpu.py -q -m gpu -k "through_lds" -p no:cacheprovider -x 2>&1 | tail -15 > $O/tests_attn_lds.txt
cat $O/tests_attn_lds.txt
{
timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
ALDM_ATTN_SCHED=3 timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
ALDM_MMA=bf16x3 timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
ALDM_MMA=bf16x3 ALDM_ATTN_SCHED=3 timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
} > $O/attn_probe.txt 2>&1; cat $O/attn_probe.txt
{
for i in 1 2; do
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/sched 1 (default): /'
ALDM_ATTN_SCHED=3 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/sched 3 (K, V^T through LDS): /'
done
} > $O/step_ab_attn_lds.txt 2>&1; cat $O/step_ab_attn_lds.txt
