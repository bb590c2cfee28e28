#!/bin/bash
# round 4, call 6: the re-parametrised model / config parity tests (both product modes, batch-8 fixtures) and the new bench line
mkdir -p gpurun_out/r4/run6
O=gpurun_out/r4/run6
rm -f gpurun_out/parity_report.txt
( time timeout 1500 python -m pytest tests/test_model_gpu.py tests/test_parity_configs_gpu.py -q -m gpu -x ) > $O/model_tests.log 2>&1; echo "model tests rc=$?"; tail -6 $O/model_tests.log | cut -c1-200
cp gpurun_out/parity_report.txt $O/parity_report.txt
( time timeout 900 python bench.py --steps 2 --warmup 1 ) > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err | cut -c1-300; cut -c1-1500 $O/bench.json
