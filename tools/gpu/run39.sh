set -x
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/gpu_suite_bx.log 2>&1; echo "gpu suite rc=$?"; tail -4 gpurun_out/gpu_suite_bx.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
