#!/bin/bash
# round 6, call 25: queries per wave of the self-attention per UNet level in f16x3 and bf16x6 (the 64-queries-per-wave rule dates from the
# round-4 six-product kernel); rows per block of the operand-stationary kernel in the 2-part mode
O=gpurun_out/r6_25; mkdir -p $O; export TMPDIR=/tmp
for MODE in f16x3 bf16x6; do
for QT in auto 1 2; do
if [ $QT = auto ]; then timeout 300 python tools/f16_attn_probe.py $MODE; else ALDM_ATTN_QT=$QT timeout 300 python tools/f16_attn_probe.py $MODE; fi
done
done 2>&1 | grep -v amdgpu.ids | tee $O/attn_qt_probe.txt
timeout 900 python tools/os_probe.py bf16x3 --rows 2>&1 | grep -v amdgpu.ids | tee $O/os_probe_rows_bf16x3.txt
