set -x
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -f csv -d /tmp/prof_trace -- python $R/bench.py --steps 1 --warmup 0 --ddim-steps 12 --no-cpu-baseline --no-roofline --no-step-probe > $R/gpurun_out/prof_trace.log 2>&1
cd $R
python tools/trace_by_grid.py /tmp/prof_trace 90 > gpurun_out/trace_by_grid.txt 2>&1
head -60 gpurun_out/trace_by_grid.txt
