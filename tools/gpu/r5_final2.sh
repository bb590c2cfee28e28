#!/bin/bash
# round 5, final call 2: HBM traffic of the igemm kernels over UNet passes only (tools/pmc_unet_pass.py; the first collection mixed the
# dominant instantiation's UNet launches with the decoder's), both modes, on the final kernel sources -> profiles/r05_pmc_traffic_<mode>.json;
# then the driver's bench command again (the line now finds its traffic entry)
set -x
O=gpurun_out/r5_final2; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for MODE in bf16x6 bf16x3; do
ALDM_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_fetch_$MODE -- python $R/tools/pmc_unet_pass.py $MODE < /dev/null > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_write_$MODE -- python $R/tools/pmc_unet_pass.py $MODE < /dev/null > /dev/null 2>&1
( cd $R && timeout -k 5 90 python tools/pmc_traffic.py /tmp/pmc_fetch_$MODE /tmp/pmc_write_$MODE $O/pmc_traffic_$MODE.json "ALDM_NO_GRAPH=1 python tools/pmc_unet_pass.py $MODE (two eager UNet passes, batch 8 x CFG)" < /dev/null > $O/pmc_traffic_$MODE.log 2>&1; tail -2 $O/pmc_traffic_$MODE.log; cp $O/pmc_traffic_$MODE.json profiles/r05_pmc_traffic_$MODE.json )
done
cd $R
( time timeout -k 5 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null ) > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; tail -4 $O/bench_final.err; cut -c1-1500 $O/bench_final.json
