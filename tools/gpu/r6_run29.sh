#!/bin/bash
# round 6, call 29: two LDS bank-conflict fixes (bitwise-neutral), as a variant library built from a patched copy of the headers:
# (a) OS kernel plain epilogue: the staged 16x16 tile with its 4x4 row index transposed (every accumulator ds_write_b32 was a 4-way conflict);
# (b) halo-patch conv: the zero-padding lanes' fragment reads go to the zero slot of their OWN 16-byte bank group instead of one shared slot
# (r06_pmc_sq_bf16x6.txt: 14.5 % LDS bank-conflict cycles in the halo kernels, 5-7 % in the OS ones, 1 % in the classic kernel).  Per-launch probe, step A/B in both fp32-grade modes, and
# the LDS counters of the halo kernels of a UNet pass on both libraries.
O=gpurun_out/r6_29; mkdir -p $O; export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/tools/gpu/libaldm_zbank.so
for L in release zbank; do
[ $L = zbank ] && export ALDM_LIB_PATH=$V || unset ALDM_LIB_PATH
timeout 300 python tools/halo_probe.py bf16x6 --quick 2>&1 | grep -v amdgpu.ids | sed "s/^/$L: /"
done > $O/halo_probe_zbank.txt 2>&1; cat $O/halo_probe_zbank.txt
{
for i in 1 2; do
for MODE in bf16x6 f16x3; do
unset ALDM_LIB_PATH
ALDM_MMA=$MODE timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed "s/^/$MODE release: /"
ALDM_LIB_PATH=$V ALDM_MMA=$MODE timeout 600 python tools/step_probe.py audioldm2-full 2 2>&1 | grep "unet step\|Error" | sed "s/^/$MODE zero slot per bank group: /"
done
done
} > $O/step_ab_zbank.txt 2>&1; cat $O/step_ab_zbank.txt
ALDM_LIB_PATH=$V timeout 900 python -m pytest tests/test_dma_gpu.py -q -m gpu -k "halo or os" 2>&1 | tail -2 | tee $O/halo_tests_zbank.txt
cd /tmp
for L in release zbank; do
[ $L = zbank ] && export ALDM_LIB_PATH=$V || unset ALDM_LIB_PATH
ALDM_NO_GRAPH=1 timeout -k 5 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --kernel-include-regex "igemm_dma_halo|igemm_dma_os" -f csv -d /tmp/pmc_lds_$L -- python $GRAFT_REPO_ROOT/tools/pmc_unet_pass.py bf16x6 < /dev/null > /dev/null 2>&1
python - $L <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(f"/tmp/pmc_lds_{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    print(f"{sys.argv[1]:8s} {k:72s} conflict/idx_active {100 * c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1):5.1f} %  "
          f"mfma busy/busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / max(c['SQ_BUSY_CYCLES'], 1):.3f}")
PY
done > $GRAFT_REPO_ROOT/$O/pmc_lds_zbank.txt 2>&1; cat $GRAFT_REPO_ROOT/$O/pmc_lds_zbank.txt
