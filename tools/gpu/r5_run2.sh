#!/bin/bash
# round 5, call 2: the operand-stationary kernel with its epilogue form as a template parameter + the residual prefetched at the top of
# the stage + the 16-instruction GELU: tests, per-shape probe (tools/os_probe.py) and same-box step A/B against the library of call 1
# (tools/gpu/libaldm_r5a.so: new attention + GEGLU lane packing, round-4 epilogue structure); then the model-level tests under the
# per-mode bars with every measured error logged
O=gpurun_out/r5_2; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_dma_gpu.py tests/test_ops_gpu.py -q -m gpu -x -p no:cacheprovider -k "os or geglu or gelu or five_product or qkv" 2>&1 | tail -5 | tee $O/tests_os.txt
{
ALDM_LIB_PATH=tools/gpu/libaldm_r5a.so timeout 600 python tools/os_probe.py bf16x6 2>&1 | grep -v amdgpu.ids | sed 's/^/call-1 library: /'
timeout 600 python tools/os_probe.py bf16x6 2>&1 | grep -v amdgpu.ids | sed 's/^/EPI template + residual prefetch: /'
} > $O/os_probe.txt 2>&1; cat $O/os_probe.txt
{
for i in 1 2; do
ALDM_LIB_PATH=tools/gpu/libaldm_r5a.so timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/call-1 library: /'
timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step" | sed 's/^/EPI template + residual prefetch + 16-op GELU: /'
done
} > $O/step_ab.txt 2>&1; cat $O/step_ab.txt
rm -f $O/err_log.tsv
ALDM_ERR_LOG=$O/err_log.tsv timeout 1500 python -m pytest tests/test_model_gpu.py -q -m gpu -p no:cacheprovider -k "unet or vae or hifigan or e2e_5step or callback or 200step" 2>&1 | tail -15 | tee $O/tests_model.txt
