#!/bin/bash
# round 3, call 28: the whole GPU suite + smoke on the committed end-of-round tree (after the last experiments were reverted)
mkdir -p gpurun_out/r3
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests/ -q -m gpu -x > gpurun_out/r3/gpu_suite_last.log 2>&1; echo "gpu suite rc=$?"; tail -3 gpurun_out/r3/gpu_suite_last.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
