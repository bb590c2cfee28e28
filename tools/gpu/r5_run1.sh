#!/bin/bash
# round 5, call 1: (a) the re-scheduled pre-split attention kernel (csrc/attn.hip, attention_d32_presplit2_kernel) against the round-4
# pipelined kernel and three build variants of itself, per launch (tools/attn_probe.py) and in the UNet step (tools/step_probe.py);
# (b) the GEGLU lane packing of the operand-stationary kernel, step A/B; (c) the per-XCC barrier price (tools/gpu/xcc_barrier.hip);
# (d) the op / DMA / attention tests under the new per-mode bars with every measured error logged (tests/tolerances.py log_err)
O=gpurun_out/r5_1; mkdir -p $O; export TMPDIR=/tmp
{
ALDM_ATTN_SCHED=0 timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
for v in attn_noslp attn_mv0 attn_mv4; do ALDM_LIB_PATH=tools/gpu/libaldm_$v.so timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids; done
ALDM_MMA=bf16x3 ALDM_ATTN_SCHED=0 timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
ALDM_MMA=bf16x3 timeout 300 python tools/attn_probe.py 2>&1 | grep -v amdgpu.ids
} > $O/attn_probe.txt 2>&1
cat $O/attn_probe.txt
timeout 120 tools/gpu/xcc_barrier 2000 > $O/xcc_barrier.txt 2>&1; cat $O/xcc_barrier.txt
{
for i in 1 2; do
ALDM_ATTN_SCHED=0 ALDM_LIB_PATH=tools/gpu/libaldm_geglu_nopack.so timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/round-4 attention + geglu: /'
ALDM_ATTN_SCHED=0 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/round-4 attention, packed geglu: /'
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/new attention, packed geglu: /'
done
ALDM_LIB_PATH=tools/gpu/libaldm_attn_noslp.so timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/new attention (no SLP build), packed geglu: /'
} > $O/step_ab.txt 2>&1
cat $O/step_ab.txt
rm -f $O/err_log.tsv
ALDM_ERR_LOG=$O/err_log.tsv timeout 1500 python -m pytest tests/test_dma_gpu.py tests/test_ops_gpu.py tests/test_parity_configs_gpu.py -q -m gpu -k "not e2e" -p no:cacheprovider 2>&1 | tail -40 > $O/tests_ops.txt
cat $O/tests_ops.txt
