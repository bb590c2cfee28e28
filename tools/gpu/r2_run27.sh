#!/bin/bash
mkdir -p gpurun_out/r2
rm -f gpurun_out/r2/bench_other.jsonl
for m in audioldm_48k audioldm2-speech-gigaspeech audioldm2-full-large-1150k; do
  timeout 600 python bench.py --model $m --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null >> gpurun_out/r2/bench_other.jsonl
done
python - <<'PY'
import json
for l in open('gpurun_out/r2/bench_other.jsonl'):
    d=json.loads(l); print(d['config']['workload'][:40], d['value'], d['unet_step_ms'])
PY
