#!/bin/bash
# Build a second library of the same ABI with gemm_w4.hip taken from another commit (same-box A/B of a kernel change):
#   tools/build_ab_lib.sh <commit> [file.hip ...]   ->  moondream_amd/libmoondream_hip_ab.so   (select with MD_HIP_LIB=<path>)
# (default file: gemm_w4.hip)
set -e
cd "$(dirname "$0")/.."
REV=${1:?commit}
D=moondream_amd/build_ab; rm -rf $D; mkdir -p $D/csrc
cp moondream_amd/csrc/*.hip moondream_amd/csrc/*.hpp $D/csrc/
shift
FILES=${@:-gemm_w4.hip}
for f in $FILES; do git show $REV:moondream_amd/csrc/$f > $D/csrc/$f; done
# entry points the current dispatcher expects from gemm_w4.hip that older revisions lack
grep -q md_gemm_w4_residual_max_cols $D/csrc/gemm_w4.hip || echo 'int md_gemm_w4_residual_max_cols() { return 1 << 30; }' >> $D/csrc/gemm_w4.hip
sed -i 's|#include "../../include/moondream_hip.h"|#include "'$PWD'/include/moondream_hip.h"|' $D/csrc/md_common.hpp
objs=""
for f in gemm_bf16 gemm_w4 gemm_fp8w attention elementwise sampling_region api; do
  extra="-mllvm -amdgpu-mfma-vgpr-form"; [ $f = gemm_w4 ] && extra=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast $extra -c $D/csrc/$f.hip -o $D/$f.o &
  objs="$objs $D/$f.o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o moondream_amd/libmoondream_hip_ab.so $objs
ls -la moondream_amd/libmoondream_hip_ab.so
