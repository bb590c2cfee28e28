#!/bin/bash
# round 6, call 17: HiFi-GAN's 64- / 32-channel stages on the DMA-fed path (A/B of Generator.DMA_MIN_CHANNELS); decode step with one bookkeeping launch less
O=gpurun_out/r6_17; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python tools/hifigan_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/hifigan_probe.txt
timeout 900 python tools/hifigan_probe.py f16x3 2>&1 | grep -v amdgpu.ids | tee -a $O/hifigan_probe.txt
PROBE_B=8 PROBE_MODES=split,split timeout 600 python tools/decode_probe.py 512 2>&1 | grep -v amdgpu.ids | tee $O/decode_probe.txt
timeout 900 python -m pytest tests/test_seqgen_gpu.py -q -m gpu -p no:cacheprovider -k "generator" 2>&1 | tail -2
