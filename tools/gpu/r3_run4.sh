#!/bin/bash
# round 3, call 4: device-side step index + drawer thread + bench.py strict re-run — model parity, env strict-mode test, host noise cost
mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_reference_binding.py -x -q > gpurun_out/r3/model_tests.log 2>&1; echo "model tests rc=$?"; tail -4 gpurun_out/r3/model_tests.log | cut -c1-300
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "env_selected or attention" > gpurun_out/r3/ops_attn_tests.log 2>&1; echo "ops attn rc=$?"; tail -3 gpurun_out/r3/ops_attn_tests.log | cut -c1-300
timeout 300 python tools/host_noise_cost.py > gpurun_out/r3/host_noise_cost.txt 2>&1; cat gpurun_out/r3/host_noise_cost.txt
timeout 900 python bench.py --steps 1 --warmup 1 --ddim-steps 20 --strict-steps 1 --no-cpu-baseline > gpurun_out/r3/bench_short.json 2> gpurun_out/r3/bench_short.err; echo "bench rc=$?"; tail -3 gpurun_out/r3/bench_short.err; cut -c1-3000 gpurun_out/r3/bench_short.json
