#!/bin/bash
# round 6, call 2: the halo kernel after its loop became branch free with the address work in front of the barrier — tests, the
# per-launch probe, and what bounds it: ablation builds of igemm_dma_halo.hip (tools/gpu/build_unit_variant.sh)
O=gpurun_out/r6_2; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_dma_gpu.py -q -m gpu -k "halo" -p no:cacheprovider -x 2>&1 | tail -5 > $O/tests_halo.txt
cat $O/tests_halo.txt
timeout 900 python tools/halo_probe.py bf16x6 2>&1 | grep -v amdgpu.ids > $O/halo_probe_bf16x6.txt; cat $O/halo_probe_bf16x6.txt
{
echo "## shipped"; timeout 300 python tools/halo_probe.py bf16x6 --quick 2>&1 | grep -v "amdgpu.ids\|^#"
for v in nopatch nob nodma noaddr nomfma noread noepi mfmaonly; do
echo "## $v"; ALDM_LIB_PATH=tools/gpu/libaldm_halo_$v.so timeout 300 python tools/halo_probe.py bf16x6 --quick 2>&1 | grep -v "amdgpu.ids\|^#"
done
} > $O/halo_ablate.txt 2>&1; cat $O/halo_ablate.txt
