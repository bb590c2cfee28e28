set -x
mkdir -p gpurun_out
ALDM_LIB_PATH=$GRAFT_REPO_ROOT/tools/gpu/libaldm_trace.so timeout 600 python tools/igemm_trace.py > gpurun_out/igemm_trace.txt 2>&1
cat gpurun_out/igemm_trace.txt
