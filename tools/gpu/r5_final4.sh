#!/bin/bash
# round 5, final call 4: the full GPU suite on the final tree (kernel sources unchanged since r5_final.sh; pipeline.py now reproduces the
# reference's second-call rand(1); the parity script's hip stage runs in the suite), every asserted error logged; smoke
O=gpurun_out/r5_final4; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -f gpurun_out/parity_report.txt $O/err_log.tsv
( time ALDM_ERR_LOG=$R/$O/err_log.tsv timeout -k 5 2400 python -m pytest tests/ -q -m gpu < /dev/null ) > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -6 $O/gpu_suite.log | cut -c1-300
cp gpurun_out/parity_report.txt $O/parity_report.txt
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -1 | tee $O/smoke.txt
