mkdir -p gpurun_out
(ALDM_LIB_PATH=tools/gpu/libaldm_pre_gn.so timeout 60 python tools/step_probe.py 2>&1 | grep "unet step"
timeout 60 python tools/step_probe.py 2>&1 | grep "unet step") | tee gpurun_out/gn_ab.txt
