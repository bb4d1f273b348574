#!/bin/bash
# Build a second library of the same ABI from the working tree with a patch applied (same-box A/B of an unmerged change):
#   tools/build_ab_patch.sh tools/patches/<name>.patch   ->  moondream_amd/libmoondream_hip_ab.so   (select with MD_HIP_LIB=<path>)
set -e
cd "$(dirname "$0")/.."
P=$(realpath ${1:?patch})
D=moondream_amd/build_ab; rm -rf $D; mkdir -p $D/moondream_amd/csrc
cp moondream_amd/csrc/*.hip moondream_amd/csrc/*.hpp $D/moondream_amd/csrc/
(cd $D && patch -p1 -s < $P)
sed -i 's|#include "../../include/moondream_hip.h"|#include "'$PWD'/include/moondream_hip.h"|' $D/moondream_amd/csrc/md_common.hpp
objs=""
for f in $(python -c "from moondream_amd import _lib; print(' '.join(s[:-4] for s in _lib.SOURCES))"); do
  extra="-mllvm -amdgpu-mfma-vgpr-form"; [ $f = gemm_w4 ] && extra=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $extra -c $D/moondream_amd/csrc/$f.hip -o $D/$f.o &
  objs="$objs $D/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o moondream_amd/libmoondream_hip_ab.so $objs
ls -la moondream_amd/libmoondream_hip_ab.so
