#!/bin/bash
# Decode step: op + generator tests, per-kernel statistics of 48 graph-replayed tokens, 512-token generation fast vs general.
set -x
O=gpurun_out/r4/run8
mkdir -p $O
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 5 200 python -m pytest tests/test_seqgen_gpu.py -q -m gpu -x < /dev/null 2>&1 | tail -4 | tee $O/tests.log
PROBE_B=8 PROBE_MODES=fast timeout -k 5 150 rocprofv3 --kernel-trace --stats -d /tmp/prof_dec -o dec --output-format csv -- python tools/decode_probe.py 48 < /dev/null > $O/prof.log 2>&1
F=$(find /tmp/prof_dec -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$F" ]; then head -8 "$F" | cut -c1-200 | tee $O/decode_kernel_stats.txt; else tail -5 $O/prof.log; fi
PROBE_MODES=general,fast,general,fast timeout -k 5 200 python tools/decode_probe.py < /dev/null 2>&1 | grep -v amdgpu.ids | tee $O/decode_probe.txt
