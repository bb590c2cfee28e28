#!/bin/bash
# round 3, call 12: the full default bench line (strict + configs) once, to see its wall time and the other configurations untuned
mkdir -p gpurun_out/r3
START=$(date +%s)
timeout 1500 python bench.py --steps 3 --warmup 1 > gpurun_out/r3/bench_full_check.json 2> gpurun_out/r3/bench_full_check.err; echo "bench rc=$? wall $(( $(date +%s) - START )) s"; tail -2 gpurun_out/r3/bench_full_check.err | cut -c1-300
python - <<'P'
import json
d=json.loads(open('gpurun_out/r3/bench_full_check.json').read().strip().splitlines()[-1])
print('value', d['value'], 'unet_step_ms', d['unet_step_ms'], 'frac', d.get('unet_step_frac_of_bf16x3_peak'))
print('strict', {k:v for k,v in d['strict'].items() if k not in ('roofline','mma','dtype')})
for k,v in d['configs'].items(): print(k, {a:b for a,b in v.items() if a not in ('roofline_tail',)}, {a:b['ms'] for a,b in v.get('roofline_tail',{}).items()})
print('tail', {a:b['ms'] for a,b in d['roofline_tail'].items()})
print('attention', d['roofline']['attention']['dominant'], d['roofline']['attention']['frac'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
P
