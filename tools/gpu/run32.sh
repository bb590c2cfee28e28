set -x
mkdir -p gpurun_out
ALDM_IGEMM_W8=1 timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "igemm or conv or linear or geglu" > gpurun_out/w8_ops.log 2>&1; echo "ops rc=$?"; tail -3 gpurun_out/w8_ops.log
for w in 0 1 0 1; do
  ALDM_IGEMM_W8=$w timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/w8_bench_$w.json
  python - <<PY
import json
d=json.load(open("gpurun_out/w8_bench_$w.json"))
print("W8=$w", d["value"], d["ms_per_step"], {k:v for k,v in d.items() if "unet" in k or "step" in k})
PY
done
cd /tmp && export TMPDIR=/tmp
ALDM_IGEMM_W8=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_w8 -o w8 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --ddim-steps 20 --no-cpu-baseline --no-roofline --no-step-probe > /dev/null 2>&1
f=$(find /tmp/prof_w8 -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/w8_kernel_stats.csv; head -25 "$f" | cut -c1-200
