#!/bin/bash
# End-of-round run on the final round-5 sources: PMC traffic of the igemm kernels in both product modes (-> profiles/r05_pmc_traffic_<mode>.json,
# stamped with the hash of the kernel sources; bench.py reports them while the hash matches), the driver's bench command with its wall time,
# rocprofv3 kernel statistics + per-grid trace + dispatch count of a 10-step job and the SQ counter table (fp32-grade mode = the headline),
# the attention / operand-stationary probes, then the full GPU suite (every measured error logged) and smoke.  Counter passes carry no
# trace domain besides --kernel-trace.
set -x
O=gpurun_out/r5_final
mkdir -p $O
R=$GRAFT_REPO_ROOT
Q="--steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-configs --no-conditioners --no-api-default"
cd /tmp; export TMPDIR=/tmp
for MODE in bf16x6 bf16x3; do
ALDM_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_fetch_$MODE -- python $R/bench.py --mma $MODE $Q < /dev/null > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "igemm" -f csv -d /tmp/pmc_write_$MODE -- python $R/bench.py --mma $MODE $Q < /dev/null > /dev/null 2>&1
( cd $R && timeout -k 5 90 python tools/pmc_traffic.py /tmp/pmc_fetch_$MODE /tmp/pmc_write_$MODE $O/pmc_traffic_$MODE.json < /dev/null > $O/pmc_traffic_$MODE.log 2>&1; tail -2 $O/pmc_traffic_$MODE.log; cp $O/pmc_traffic_$MODE.json profiles/r05_pmc_traffic_$MODE.json )
done
cd $R
( time timeout -k 5 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null ) > $O/bench_final.json 2> $O/bench_final.err; echo "bench rc=$?"; tail -4 $O/bench_final.err; cut -c1-1500 $O/bench_final.json
cd /tmp
timeout -k 5 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_fin -o fin --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-roofline --no-step-probe --no-fast --no-configs --no-conditioners --no-api-default < /dev/null > /dev/null 2>&1
cd $R
cp $(find /tmp/prof_fin -name "*kernel_stats.csv" | head -1) $O/kernel_stats_bf16x6.csv
mkdir -p /tmp/kt && cp $(find /tmp/prof_fin -name "*kernel_trace.csv" | head -1) /tmp/kt/ && python tools/trace_by_grid.py /tmp/kt 100 > $O/trace_by_grid_bf16x6.txt 2>&1
python tools/trace_copy_attrib.py /tmp/kt 10 > $O/trace_copy_attrib.txt 2>&1; cat $O/trace_copy_attrib.txt
head -12 $O/kernel_stats_bf16x6.csv | cut -c1-150
cd /tmp
KRE="igemm_dma|attention|layernorm|gn_partial|split_rows"
ALDM_NO_GRAPH=1 timeout -k 5 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq1 -- python $R/bench.py $Q < /dev/null > /dev/null 2>&1
ALDM_NO_GRAPH=1 timeout -k 5 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --kernel-include-regex "$KRE" -f csv -d /tmp/pmc_sq2 -- python $R/bench.py $Q < /dev/null > /dev/null 2>&1
cd $R
python tools/pmc_sq_table.py /tmp/pmc_sq1 /tmp/pmc_sq2 > $O/pmc_sq_bf16x6.txt 2>&1; head -26 $O/pmc_sq_bf16x6.txt
timeout -k 5 200 python tools/attn_probe.py < /dev/null 2>&1 | grep -v amdgpu.ids | tee $O/attn_probe.txt
timeout -k 5 300 python tools/os_probe.py bf16x6 < /dev/null 2>&1 | grep -v amdgpu.ids | tee $O/os_probe.txt
rm -f gpurun_out/parity_report.txt $O/err_log.tsv
( time ALDM_ERR_LOG=$R/$O/err_log.tsv timeout -k 5 2400 python -m pytest tests/ -q -m gpu < /dev/null ) > $O/gpu_suite.log 2>&1; echo "gpu suite rc=$?"; tail -5 $O/gpu_suite.log | cut -c1-200
cp gpurun_out/parity_report.txt $O/parity_report.txt
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null 2>&1 | tail -1 | tee $O/smoke.txt
