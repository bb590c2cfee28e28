#!/bin/bash
# round 6: the full GPU suite with every asserted error logged, then smoke
O=gpurun_out/r6_suite; mkdir -p $O; export TMPDIR=/tmp
rm -f $O/err_log.tsv
ALDM_ERR_LOG=$O/err_log.tsv timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -40 > $O/gpu_suite.log
cat $O/gpu_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -3 $O/smoke.txt
