#!/bin/bash
# round 5, final call 5: the tree with the row-maximum fix in the pipelined / pre-split attention kernels (default kernels' numerics
# change at rounding level; igemm sources untouched, so the PMC traffic records stand): the full GPU suite with every asserted error
# logged, smoke, and the bench line
O=gpurun_out/r5_final5; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
rm -f gpurun_out/parity_report.txt $O/err_log.tsv
timeout 600 python -m pytest tests/test_dma_gpu.py -q -m gpu -p no:cacheprovider -k "attention or presplit" < /dev/null 2>&1 | tail -6 | cut -c1-300 | tee $O/tests_attn.txt
( time ALDM_ERR_LOG=$R/$O/err_log.tsv timeout -k 5 1500 python -m pytest tests/ -q -m gpu < /dev/null ) > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -6 $O/gpu_suite.log | cut -c1-300
cp gpurun_out/parity_report.txt $O/parity_report.txt
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -1 | tee $O/smoke.txt
( time timeout -k 5 600 python3 bench.py --gpus 1 --steps 5 --warmup 2 < /dev/null ) > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; tail -4 $O/bench_final.err
python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r5_final5/bench_final.json").read().splitlines() if l.startswith("{")][-1])
print(d["value"], d["unet_step_ms"], d["roofline"]["frac"], d["roofline"].get("traffic_over_algorithmic"), d["roofline"].get("attention", {}), d.get("replicas_one_gpu"))
PY
