#!/bin/bash
# round 6, final call 2: the driver's bench command once more on the final sources (another box: the spread; bench.py's roofline block
# with per-instantiation fractions)
O=gpurun_out/r6_final2; mkdir -p $O; export TMPDIR=/tmp
( time timeout -k 5 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null ) > $O/bench_final_8.json 2> $O/bench_final_8.err; echo "bench rc=$?"; tail -4 $O/bench_final_8.err; cut -c1-600 $O/bench_final_8.json
