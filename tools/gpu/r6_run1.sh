#!/bin/bash
# round 6, call 1: the halo-patch 3x3 convolution kernel — its GPU tests, then per-launch times against the tuned kernels
O=gpurun_out/r6_1; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_dma_gpu.py -q -m gpu -k "halo or five_product" -p no:cacheprovider -x 2>&1 | tail -30 > $O/tests_halo.txt
cat $O/tests_halo.txt
timeout 900 python tools/halo_probe.py bf16x6 2>&1 | grep -v amdgpu.ids > $O/halo_probe_bf16x6.txt; cat $O/halo_probe_bf16x6.txt
timeout 600 python tools/halo_probe.py bf16x6 --vae 2>&1 | grep -v amdgpu.ids > $O/halo_probe_vae_bf16x6.txt; cat $O/halo_probe_vae_bf16x6.txt
