mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu > gpurun_out/bx_ops3.log 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/bx_ops3.log
timeout 900 python tools/igemm_autotune.py --mma bf16x6 gpurun_out/mi355x_igemm_bf16x6.json audioldm2-full audioldm_48k audioldm2-full-large-1150k audioldm2-speech-gigaspeech > gpurun_out/autotune_bx.txt 2>&1; echo "autotune rc=$?"; tail -4 gpurun_out/autotune_bx.txt | cut -c1-300
