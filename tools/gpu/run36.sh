mkdir -p gpurun_out/pmc
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
run() { # name, counters, args...
  n=$1; c=$2; shift 2
  timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$n -o $n --output-format csv -- python $R/tools/pmc_one.py "$@" > /dev/null 2>&1
  f=$(find /tmp/pmc_$n -name "*counter_collection.csv" | head -1)
  python - "$f" "$n" <<'PY'
import csv, sys, collections
f, n = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "igemm_kernel" not in k: continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(n, k[:70], {c: round(v / cnt[(k, c)]) for c, v in d.items()})
PY
}
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS"
run bx128a "$SQ1" bf16x6 128 128 2>&1 | tee -a $R/gpurun_out/pmc/summary.txt
run bx128b "$SQ2" bf16x6 128 128 2>&1 | tee -a $R/gpurun_out/pmc/summary.txt
run f128a "$SQ1" f32 128 128 2>&1 | tee -a $R/gpurun_out/pmc/summary.txt
run f128b "$SQ2" f32 128 128 2>&1 | tee -a $R/gpurun_out/pmc/summary.txt
