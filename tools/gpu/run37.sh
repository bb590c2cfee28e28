mkdir -p gpurun_out
export ALDM_MMA=bf16x6
timeout 600 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/bx_bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bx_bench.json"))
print("BX bench", d["value"], d["ms_per_step"], {k:v for k,v in d.items() if "unet" in k})
PY
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "e2e or unet_full or vae or hifigan" > gpurun_out/bx_model.log 2>&1; echo "model rc=$?"; tail -5 gpurun_out/bx_model.log
cp gpurun_out/parity_report.txt gpurun_out/bx_parity_report.txt 2>/dev/null; cat gpurun_out/bx_parity_report.txt 2>/dev/null | cut -c1-250
