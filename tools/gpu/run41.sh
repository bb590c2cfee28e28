mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 200 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "pytorch_rocm_eager" 2>&1 | tail -1
cat gpurun_out/parity_report.txt | cut -c1-200
rm -f gpurun_out/bench_other_bx.jsonl
for m in audioldm_48k audioldm2-speech-gigaspeech audioldm2-full-large-1150k; do
  timeout 150 python bench.py --model $m --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 >> gpurun_out/bench_other_bx.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/bench_other_bx.jsonl"):
    try:
        d = json.loads(l)
    except Exception:
        print("bad line", l[:80]); continue
    print(d["config"]["workload"][:28], d["value"], d["ms_per_step"], d.get("unet_step_ms"), d.get("unet_step_frac_of_f32_mfma_peak"), d["roofline"]["achieved"], d["roofline"]["kernel"])
PY
