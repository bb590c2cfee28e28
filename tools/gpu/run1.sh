set -x
mkdir -p gpurun_out
python tools/unet_shapes.py 8 > gpurun_out/unet_shapes_v0.txt 2> gpurun_out/unet_shapes_v0.err
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/counters.txt 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $R/gpurun_out/pmc1 -- python $R/tools/igemm_micro.py 3 > $R/gpurun_out/pmc1.log 2>&1 || true
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $R/gpurun_out/pmc2 -- python $R/tools/igemm_micro.py 3 > $R/gpurun_out/pmc2.log 2>&1 || true
rocprofv3 --kernel-trace --pmc WRITE_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_VMEM -f csv -d $R/gpurun_out/pmc3 -- python $R/tools/igemm_micro.py 3 > $R/gpurun_out/pmc3.log 2>&1 || true
cd $R
for d in pmc1 pmc2 pmc3; do python tools/pmc_table.py gpurun_out/$d > gpurun_out/$d.txt 2>&1; find gpurun_out/$d -name "*.csv" -size +2M -delete; done
ls -R gpurun_out | head -50
