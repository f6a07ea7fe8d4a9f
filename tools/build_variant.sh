#!/bin/bash
# A variant of libpyg_hip.so that differs in ONE translation unit's flags:  tools/build_variant.sh <name> <unit, e.g. rgcn> <flags ...>
# -> pyg_lib_amd/libpyg_hip_<name>.so (git-ignored; travels with the gpurun snapshot; a lease script copies it over libpyg_hip.so)
set -e
cd "$(dirname "$0")/../pyg_lib_amd/csrc"
name=$1; unit=$2; shift 2
extra=""
case "$unit" in rgcn) extra="-mllvm -amdgpu-mfma-vgpr-form=1" ;; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -fvisibility=hidden -I../../include -Ihip $extra "$@" -c hip/$unit.hip -o build/${unit}_$name.o
objs=""
for f in hip/*.hip; do b=$(basename $f .hip); if [ $b = $unit ]; then objs="$objs build/${unit}_$name.o"; else objs="$objs build/$b.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o ../libpyg_hip_$name.so
echo built libpyg_hip_$name.so
