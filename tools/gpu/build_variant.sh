#!/bin/bash
# Build a variant of libaldm_hip.so with extra compiler flags into tools/gpu/libaldm_<name>.so (for same-box A/Bs with
# ALDM_LIB_PATH=tools/gpu/libaldm_<name>.so python tools/step_probe.py / pytest).  Example:
#   tools/gpu/build_variant.sh glds -DALDM_BX_GLDS=1
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; shift
d=/tmp/variant_$name; rm -rf $d; mkdir -p $d/audioldm2_amd $d/include
cp -r $ROOT/audioldm2_amd/csrc $d/audioldm2_amd/; cp $ROOT/include/*.h $d/include/
rm -f $d/audioldm2_amd/csrc/*.o
# (EXTRA_CXXFLAGS, not CXXFLAGS: a command-line CXXFLAGS also overrides the Makefile's per-file additions — until round 5's last call
#  the variants of attn.hip were therefore built WITHOUT -amdgpu-mfma-vgpr-form=1)
make -C $d/audioldm2_amd/csrc -j8 EXTRA_CXXFLAGS="$*" > $d/build.log 2>&1 || { tail -20 $d/build.log; exit 1; }
grep -q "amdgpu-mfma-vgpr-form=1 .*-c attn.hip" $d/build.log || { echo "attn.hip was built without its per-file flags"; exit 1; }
cp $d/audioldm2_amd/libaldm_hip.so $ROOT/tools/gpu/libaldm_$name.so
echo "built tools/gpu/libaldm_$name.so with: $*"
