#!/bin/bash
# End-of-round run on the final round-6 sources: PMC traffic of the igemm kernels over UNet passes (-> profiles/r06_pmc_traffic_<mode>.json, stamped with
# the kernel-source hash; bench.py reports them while the hash matches), the driver's bench command with its wall time, rocprofv3 kernel statistics +
# per-grid trace + per-step breakdown of a 10-step job in both fp32-grade modes, the SQ counter table of the headline mode, then the full GPU suite
# (every measured error logged) and smoke.  Counter passes carry no trace domain besides --kernel-trace.
set -x
O=gpurun_out/r6_final
mkdir -p $O
R=$GRAFT_REPO_ROOT
Q="--steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-f16x3 --no-configs --no-conditioners --no-api-default"
cd /tmp; export TMPDIR=/tmp
for MODE in bf16x6 f16x3 bf16x3; do
ALDM_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_fetch_$MODE -- python $R/tools/pmc_unet_pass.py $MODE < /dev/null > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_write_$MODE -- python $R/tools/pmc_unet_pass.py $MODE < /dev/null > /dev/null 2>&1
( cd $R && timeout -k 5 90 python tools/pmc_traffic.py /tmp/pmc_fetch_$MODE /tmp/pmc_write_$MODE $O/pmc_traffic_$MODE.json "ALDM_NO_GRAPH=1 python tools/pmc_unet_pass.py $MODE (two eager UNet passes, batch 8 x CFG)" < /dev/null > $O/pmc_traffic_$MODE.log 2>&1; tail -2 $O/pmc_traffic_$MODE.log; cp $O/pmc_traffic_$MODE.json profiles/r06_pmc_traffic_$MODE.json )
done
cd $R
( time timeout -k 5 1200 python3 bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null ) > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; tail -4 $O/bench_final.err; cut -c1-1500 $O/bench_final.json
for MODE in bf16x6 f16x3; do
cd /tmp
rm -rf /tmp/prof_$MODE /tmp/kt_$MODE
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$MODE -o fin --output-format csv -- python $R/bench.py --mma $MODE --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-f16x3 --no-configs --no-conditioners --no-api-default < /dev/null > /dev/null 2>&1
cd $R
cp $(find /tmp/prof_$MODE -name "*kernel_stats.csv" | head -1) $O/kernel_stats_$MODE.csv
mkdir -p /tmp/kt_$MODE && cp $(find /tmp/prof_$MODE -name "*kernel_trace.csv" | head -1) /tmp/kt_$MODE/
python tools/trace_by_grid.py /tmp/kt_$MODE 100 > $O/trace_by_grid_$MODE.txt 2>&1
python tools/trace_step_breakdown.py /tmp/kt_$MODE 40 > $O/step_breakdown_$MODE.txt 2>&1; head -16 $O/step_breakdown_$MODE.txt
done
python tools/trace_copy_attrib.py /tmp/kt_bf16x6 10 > $O/trace_copy_attrib.txt 2>&1; cat $O/trace_copy_attrib.txt
cd /tmp
KRE="igemm_dma|attention|layernorm|gn_partial|split_rows"
ALDM_NO_GRAPH=1 timeout -k 5 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq1 -- python $R/bench.py $Q < /dev/null > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout -k 5 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq2 -- python $R/bench.py $Q < /dev/null > /dev/null 2>&1
cd $R
python tools/pmc_sq_table.py /tmp/pmc_sq1 /tmp/pmc_sq2 > $O/pmc_sq_bf16x6.txt 2>&1; head -30 $O/pmc_sq_bf16x6.txt
cd /tmp
ALDM_NO_GRAPH=1 timeout -k 5 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq1h -- python $R/bench.py --mma f16x3 $Q < /dev/null > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout -k 5 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq2h -- python $R/bench.py --mma f16x3 $Q < /dev/null > /dev/null 2>&1
cd $R
python tools/pmc_sq_table.py /tmp/pmc_sq1h /tmp/pmc_sq2h > $O/pmc_sq_f16x3.txt 2>&1; head -12 $O/pmc_sq_f16x3.txt
rm -f gpurun_out/parity_report.txt $O/err_log.tsv
( time ALDM_ERR_LOG=$R/$O/err_log.tsv timeout -k 5 3000 python -m pytest tests/ -q -m gpu < /dev/null ) > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -5 $O/gpu_suite.log | cut -c1-200
cp gpurun_out/parity_report.txt $O/parity_report.txt
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -1 | tee $O/smoke.txt
