#!/bin/bash
# round 6, call 3: is the halo kernel's weight stream really what its ablation says?  "nob" feeds ZERO weights (less matrix-pipe power,
# higher clock); "bres" re-reads weight tile 0 (real bits, cache resident), "pres" channel block 0's patch, "bpres" both.
O=gpurun_out/r6_3; mkdir -p $O; export TMPDIR=/tmp
{
echo "## shipped"; timeout 300 python tools/halo_probe.py bf16x6 --quick 2>&1 | grep -v "amdgpu.ids\|^#"
for v in nob bres pres bpres bpres_noepi mfmaonly noepi nomfma; do
echo "## $v"; ALDM_LIB_PATH=tools/gpu/libaldm_halo_$v.so timeout 300 python tools/halo_probe.py bf16x6 --quick 2>&1 | grep -v "amdgpu.ids\|^#"
done
echo "## shipped again"; timeout 300 python tools/halo_probe.py bf16x6 --quick 2>&1 | grep -v "amdgpu.ids\|^#"
} > $O/halo_ablate2.txt 2>&1; cat $O/halo_ablate2.txt
