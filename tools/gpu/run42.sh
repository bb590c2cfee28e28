mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "groupnorm or conv_fused or wave" 2>&1 | tail -1
timeout 400 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/bench_gn.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_gn.json"))
print("bench", d["value"], d["ms_per_step"], {k:v for k,v in d.items() if "unet" in k})
PY
