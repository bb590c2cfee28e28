set -x
mkdir -p gpurun_out/r2
timeout 900 python -m pytest tests/test_t5.py tests/test_clap_text.py tests/test_phoneme.py tests/test_dma_gpu.py tests/test_ops_gpu.py -q -m gpu 2>&1 | tail -8
timeout 600 python -m pytest tests/test_model_gpu.py -q -m gpu -k "rccl or two_rank" -s 2>&1 | tail -8
