#!/bin/bash
# round 3, call 8: the whole GPU suite on the new fixtures (vocoder gain, between-sample tolerances, conditioner drop-in e2e)
mkdir -p gpurun_out/r3
rm -f gpurun_out/parity_report.txt
timeout 2400 python -m pytest tests/ -q -m gpu > gpurun_out/r3/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -15 gpurun_out/r3/gpu_suite.log | cut -c1-300
cp gpurun_out/parity_report.txt gpurun_out/r3/parity_report.txt
grep -h "conditioner e2e" gpurun_out/r3/gpu_suite.log
