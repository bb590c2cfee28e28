#!/bin/bash
# round 5, call 3: the shared classifier-free-guidance prefix (UNetModel.forward cfg_shared: the context-free head of the UNet on the 8
# prompts instead of the 16 rows) — tests + same-box step A/B against ALDM_CFG_SHARE=0; deeper LDS rings of the 64x64 loader-wave tile
# on the 1024-row GEMMs (tools/lw_ring_probe.py)
O=gpurun_out/r5_3; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -x -p no:cacheprovider -k "shared_cfg or cfg_batched or e2e_5step or batch8 or callback" 2>&1 | tail -4 | tee $O/tests.txt
timeout 600 python tools/lw_ring_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/lw_ring_probe.txt
{
for i in 1 2; do
ALDM_CFG_SHARE=0 timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/16-row pass: /'
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/shared prefix: /'
done
ALDM_CFG_SHARE=0 timeout 600 python tools/step_probe.py audioldm2-full-large-1150k 2 2>&1 | grep "unet step" | sed 's/^/large, 16-row pass: /'
timeout 600 python tools/step_probe.py audioldm2-full-large-1150k 2 2>&1 | grep "unet step" | sed 's/^/large, shared prefix: /'
} > $O/step_ab.txt 2>&1; cat $O/step_ab.txt
