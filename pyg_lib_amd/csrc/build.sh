#!/bin/bash
# Build libpyg_hip.so (torch-free C-ABI library, gfx950) in-tree.
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
OUT=../libpyg_hip.so
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -fPIC -fvisibility=hidden -I../../include -Ihip $EXTRA_HIPCC_FLAGS"
mkdir -p build
objs=""
pids=""
for f in hip/*.hip; do
  o=build/$(basename "$f" .hip).o
  stale=0
  for h in hip/*.h ../../include/pyg_hip.h; do [ "$h" -nt "$o" ] && stale=1; done
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ $stale = 1 ]; then
    echo "hipcc $f"
    extra=""
    # rgcn.hip: MFMA results in VGPRs -- its accumulators are packed by VALU instructions right after the k loop, and from
    # AGPRs that costs 64 v_accvgpr_read per 32-row tile (a fifth of the kernel's VALU work) and 40 more registers
    case "$f" in hip/rgcn.hip) extra="-mllvm -amdgpu-mfma-vgpr-form=1" ;; esac
    $HIPCC $FLAGS $extra -c "$f" -o "$o" &
    pids="$pids $!"
  fi
  objs="$objs $o"
done
for p in $pids; do wait $p || { echo "hipcc failed"; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o $OUT
echo "built $OUT"
