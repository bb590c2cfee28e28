set -x
mkdir -p gpurun_out
ALDM_IGEMM_VARIANT=3 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q > gpurun_out/ops_test_v5.log 2>&1; echo "ops(variant3) rc=$?"
tail -3 gpurun_out/ops_test_v5.log
for V in 0 1 2 3; do
  ALDM_IGEMM_VARIANT=$V timeout 300 python tools/unet_shapes.py 8 > gpurun_out/unet_shapes_var$V.txt 2>/dev/null
  head -1 gpurun_out/unet_shapes_var$V.txt
  ALDM_IGEMM_VARIANT=$V timeout 300 python tools/bench_ops.py 16 2>/dev/null | grep -v "attn\|gn_\|layernorm\|device" > gpurun_out/bench_ops_var$V.txt
done
paste gpurun_out/bench_ops_var0.txt gpurun_out/bench_ops_var1.txt gpurun_out/bench_ops_var2.txt gpurun_out/bench_ops_var3.txt | awk '{print $1,$2,$3,$4, $(NF-39) "|", $5,$6, $12,$13, $19,$20, $26,$27}' | head -20
