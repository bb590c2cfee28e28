#!/bin/bash
# round 3, call 10: VAE decoder / encoder and the wide HiFi-GAN stages on the DMA-fed GEMM — parity, then the tail timings
mkdir -p gpurun_out/r3
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_dma_gpu.py -x -q > gpurun_out/r3/model_tests2.log 2>&1; echo "model tests rc=$?"; tail -4 gpurun_out/r3/model_tests2.log | cut -c1-300
grep -h "vae\|hifigan\|e2e 200\|48k" gpurun_out/parity_report.txt | cut -c1-220
timeout 900 python bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-strict --no-cpu-baseline --no-step-probe > gpurun_out/r3/bench_tail.json 2> gpurun_out/r3/bench_tail.err; echo "bench rc=$?"; tail -2 gpurun_out/r3/bench_tail.err | cut -c1-300
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3/bench_tail.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('roofline_tail')))
P
ALDM_DMA=0 timeout 900 python bench.py --steps 1 --warmup 1 --ddim-steps 4 --no-strict --no-cpu-baseline --no-step-probe --model audioldm_48k > gpurun_out/r3/bench_tail48_old.json 2>/dev/null
timeout 900 python bench.py --steps 1 --warmup 1 --ddim-steps 4 --no-strict --no-cpu-baseline --no-step-probe --model audioldm_48k > gpurun_out/r3/bench_tail48.json 2>/dev/null
python - <<'P'
import json
for f in ('gpurun_out/r3/bench_tail48_old.json','gpurun_out/r3/bench_tail48.json'):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, json.dumps({k:(v['ms'],v['achieved']) for k,v in d['roofline_tail'].items()}))
    except Exception as e: print(f, 'failed', e)
P
