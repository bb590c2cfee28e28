#!/bin/bash
# round 5, call 10: the one-pass fixed-reference pre-split attention kernel (attention_d32_presplit3_kernel) against the library of the
# final records (tools/gpu/libaldm_r5base.so = HEAD 30fd317's sources, hash 3db7952cc39fe419): per-launch probe in both product modes
# and for all three ALDM_ATTN_SCHED arms, the attention tests (late-large-score cases through the slow path included), same-box step A/B
O=gpurun_out/r5_10; mkdir -p $O; export TMPDIR=/tmp
{
ALDM_LIB_PATH=tools/gpu/libaldm_r5base.so timeout 300 python tools/attn_probe.py 2>&1 | grep " us "
ALDM_ATTN_SCHED=1 timeout 300 python tools/attn_probe.py 2>&1 | grep " us "
timeout 300 python tools/attn_probe.py 2>&1 | grep " us "
ALDM_LIB_PATH=tools/gpu/libaldm_r5base.so ALDM_MMA=bf16x3 timeout 300 python tools/attn_probe.py 2>&1 | grep " us "
ALDM_MMA=bf16x3 timeout 300 python tools/attn_probe.py 2>&1 | grep " us "
} > $O/attn_probe.txt 2>&1; cat $O/attn_probe.txt
rm -f $O/err_log.tsv
ALDM_ERR_LOG=$GRAFT_REPO_ROOT/$O/err_log.tsv timeout 900 python -m pytest tests/test_dma_gpu.py tests/test_ops_gpu.py tests/test_parity_configs_gpu.py -q -m gpu -p no:cacheprovider -k "attention or presplit or qkv or attn" 2>&1 | tail -12 | tee $O/tests_attn.txt
{
for i in 1 2; do
ALDM_LIB_PATH=tools/gpu/libaldm_r5base.so timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/final-record library: /'
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/one-pass attention: /'
done
} > $O/step_ab.txt 2>&1; cat $O/step_ab.txt
grep -i "attn\|attention\|presplit" $O/err_log.tsv | sort -t$'\t' -k2 -g | tail -12
