#!/bin/bash
# Build an experimental variant of libaurora_hip.so:  tools/build_variant.sh <name> [extra hipcc flags for gemm.hip ...]
# -> aurora_amd/_lib/libaurora_hip_<name>.so   (select with AURORA_HIP_LIB=<path>)
set -e
NAME=$1; shift
D=aurora_amd/_lib/var_$NAME; mkdir -p $D
CC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1"
for f in runtime attention norm embed band model step; do
  if [ ! -f aurora_amd/_lib/var_cache/$f.o ] || [ aurora_amd/csrc/$f.hip -nt aurora_amd/_lib/var_cache/$f.o ]; then
    mkdir -p aurora_amd/_lib/var_cache; $CC -c aurora_amd/csrc/$f.hip -o aurora_amd/_lib/var_cache/$f.o &
  fi
done
$CC "$@" -c aurora_amd/csrc/gemm.hip -o $D/gemm.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC aurora_amd/_lib/var_cache/*.o $D/gemm.o -o aurora_amd/_lib/libaurora_hip_$NAME.so
rm -rf $D
echo built aurora_amd/_lib/libaurora_hip_$NAME.so
