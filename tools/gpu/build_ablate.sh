#!/bin/bash
# Debug builds of libaldm_hip.so with pieces of the bf16-split K loop removed (ALDM_ABLATE bit mask, see
# csrc/igemm_kernel.h) -> tools/gpu/libaldm_abl<mask>.so; timing only, results are wrong by construction.
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
for m in "$@"; do
  (
    d=/tmp/abl_$m; rm -rf $d; mkdir -p $d/audioldm2_amd $d/include
    cp -r $ROOT/audioldm2_amd/csrc $d/audioldm2_amd/; cp $ROOT/include/*.h $d/include/
    rm -f $d/audioldm2_amd/csrc/*.o
    make -C $d/audioldm2_amd/csrc -j16 CXXFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -Wno-unused-variable -DALDM_ABLATE=$m" > $d/build.log 2>&1
    cp $d/audioldm2_amd/libaldm_hip.so $ROOT/tools/gpu/libaldm_abl$m.so
    echo "built mask $m"
  ) &
done
wait
