mkdir -p gpurun_out
python tools/ablate_run.py 2>&1 | grep "|" | tee gpurun_out/ablate.txt
for m in 1 3 4 12 16 28 31; do
  ALDM_LIB_PATH=tools/gpu/libaldm_abl$m.so python tools/ablate_run.py 2>&1 | grep "|" | tee -a gpurun_out/ablate.txt
done
