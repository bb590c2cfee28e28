mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "splitk or wave_groups or deep_level" 2>&1 | tail -1
timeout 600 python bench.py > gpurun_out/bench_final2.json 2>/dev/null; cut -c1-400 gpurun_out/bench_final2.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_final2.json"))
print("bench", d["value"], d["ms_per_step"], {k:v for k,v in d.items() if "unet" in k}, d["roofline"]["achieved"], d["roofline"]["traffic"], d["roofline"]["kernel"], d["cpu_baseline"]["value"])
PY
