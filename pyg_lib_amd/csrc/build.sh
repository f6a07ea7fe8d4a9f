#!/bin/bash
# Build libpyg_hip.so (torch-free C-ABI library, gfx950) in-tree.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libpyg_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -I../../include -Ihip"
mkdir -p build
objs=""
for f in hip/*.hip; do
  o=build/$(basename "$f" .hip).o
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ hip/common.h -nt "$o" ] || [ ../../include/pyg_hip.h -nt "$o" ]; then
    echo "hipcc $f"
    $HIPCC $FLAGS -c "$f" -o "$o"
  fi
  objs="$objs $o"
done
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o $OUT
echo "built $OUT"
