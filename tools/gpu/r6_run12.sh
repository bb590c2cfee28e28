#!/bin/bash
# round 6, call 12: the driver's bench command on the mid-round tree (halo forms, f16x3 sub-record with fp16 attention)
O=gpurun_out/r6_12; mkdir -p $O; export TMPDIR=/tmp
( time timeout -k 5 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null ) > $O/bench_mid.json 2> $O/bench_mid.err; echo "bench rc=$?"; tail -4 $O/bench_mid.err; cut -c1-3000 $O/bench_mid.json
