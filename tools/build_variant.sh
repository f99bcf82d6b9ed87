#!/bin/bash
# Build an experimental variant of libaurora_hip.so:  tools/build_variant.sh <name> [extra hipcc flags for gemm.hip ...]
# -> aurora_amd/_lib/libaurora_hip_<name>.so   (select with AURORA_HIP_LIB=<path>)
set -e
NAME=$1; shift
D=aurora_amd/_lib/var_$NAME; mkdir -p $D
for f in runtime attention norm embed; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -amdgpu-mfma-vgpr-form=1 -c aurora_amd/csrc/$f.hip -o $D/$f.o &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c aurora_amd/csrc/gemm.hip -o $D/gemm.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/*.o -o aurora_amd/_lib/libaurora_hip_$NAME.so
rm -rf $D
echo built aurora_amd/_lib/libaurora_hip_$NAME.so
