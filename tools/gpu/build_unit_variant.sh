#!/bin/bash
# Build a variant of libaldm_hip.so in which ONE translation unit is recompiled with extra flags (the other objects are the
# in-tree ones — run `make -C audioldm2_amd/csrc` first) into tools/gpu/libaldm_<name>.so.  Example:
#   tools/gpu/build_unit_variant.sh halo_nomfma igemm_dma_halo -DALDM_DMA_ABLATE=4
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
name=$1; unit=$2; shift; shift
C=$ROOT/audioldm2_amd/csrc
extra=""; [ "$unit" = attn ] && extra="-mllvm -amdgpu-mfma-vgpr-form=1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function $extra "$@" -c $C/$unit.hip -o /tmp/${unit}_$name.o
objs=$(ls $C/*.o | grep -v "\.th\.o" | grep -v "/$unit\.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/${unit}_$name.o -o $ROOT/tools/gpu/libaldm_$name.so
echo "built tools/gpu/libaldm_$name.so ($unit.hip with: $*)"
