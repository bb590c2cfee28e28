#!/bin/bash
# round 5, final call 3: the driver's bench command once more on the final sources (bench.py now also reports `replicas_one_gpu`)
O=gpurun_out/r5_final3; mkdir -p $O; export TMPDIR=/tmp
( time timeout -k 5 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null ) > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; tail -4 $O/bench_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5_final3/bench_final.json").read().splitlines() if l.startswith("{")][-1])
print(d["value"], d["unet_step_ms"], d["roofline"]["frac"], d["roofline"]["traffic_over_algorithmic"], d.get("replicas_one_gpu"))
PY
