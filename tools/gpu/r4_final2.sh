#!/bin/bash
# Re-collection after csrc/decode.hip joined the library (new kernel-source hash): decode / ABI / conditioner-stack tests, PMC traffic
# of the igemm kernels in both product modes (-> profiles/r04_pmc_traffic_<mode>.json), the driver's bench command, the decode
# kernels per shape and the 512-token generation A/B, smoke.  (The full GPU suite of the round ran on the sources before decode.hip:
# profiles/r04_gpu_suite.log; nothing it covers changed.)  Every step bounded; nothing reads stdin.
set -x
O=gpurun_out/r4/final2
mkdir -p $O
R=$GRAFT_REPO_ROOT
cd $R
( time timeout -k 5 240 python -m pytest tests/test_seqgen_gpu.py tests/test_abi.py tests/test_reference_binding.py -q -m gpu < /dev/null ) > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log | cut -c1-200
Q="--steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-configs --no-conditioners --no-api-default"
cd /tmp; export TMPDIR=/tmp
for MODE in bf16x6 bf16x3; do
ALDM_NO_GRAPH=1 timeout -k 5 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_fetch_$MODE -- python $R/bench.py --mma $MODE $Q < /dev/null > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout -k 5 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_write_$MODE -- python $R/bench.py --mma $MODE $Q < /dev/null > /dev/null 2>&1
( cd $R && timeout -k 5 60 python tools/pmc_traffic.py /tmp/pmc_fetch_$MODE /tmp/pmc_write_$MODE $O/pmc_traffic_$MODE.json < /dev/null > $O/pmc_traffic_$MODE.log 2>&1; tail -2 $O/pmc_traffic_$MODE.log; cp $O/pmc_traffic_$MODE.json profiles/r04_pmc_traffic_$MODE.json )
done
cd $R
( time timeout -k 5 420 python3 bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null ) > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; tail -4 $O/bench_final.err; cut -c1-600 $O/bench_final.json
timeout -k 5 90 python tools/decode_shapes.py < /dev/null 2>&1 | grep -v amdgpu.ids | tee $O/decode_shapes.txt
timeout -k 5 150 python tools/decode_probe.py < /dev/null 2>&1 | grep -v amdgpu.ids | tee $O/decode_probe.txt
timeout -k 5 90 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -1
