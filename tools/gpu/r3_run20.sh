#!/bin/bash
# round 3, call 20: the driver's own bench command (--steps 20 --warmup 5) once, with its wall time
mkdir -p gpurun_out/r3
S=$(date +%s)
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r3/bench_driver_cmd.json 2> gpurun_out/r3/bench_driver_cmd.err; echo "rc=$?"
E=$(date +%s); echo "wall $((E-S)) s" | tee gpurun_out/r3/bench_driver_cmd.wall
cut -c1-400 gpurun_out/r3/bench_driver_cmd.json
