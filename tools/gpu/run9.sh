set -x
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v8.log 2>&1; echo "ops rc=$?"; tail -2 gpurun_out/ops_test_v8.log
for Q in 1 2; do ALDM_ATTN_QT=$Q timeout 300 python tools/bench_ops.py 16 2>/dev/null | grep attn | sed "s/^/QT=$Q /"; done
timeout 1500 python tools/igemm_autotune.py gpurun_out/mi355x_igemm.json audioldm2-full audioldm_48k > gpurun_out/autotune.log 2>&1; echo "tune rc=$?"; tail -4 gpurun_out/autotune.log
