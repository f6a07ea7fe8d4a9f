#!/bin/bash
# Build libpyg.so: the torch operator library (namespace `pyg`) on top of the C-ABI libpyg_hip.so.
set -e
cd "$(dirname "$0")"
CXX=${CXX:-g++}
OUT=../libpyg.so
TORCH=$(python -c "import torch, os; print(os.path.dirname(torch.__file__))")
ABI=$(python -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
mkdir -p build
objs=""
pids=""
for SRC in binding/*.cpp; do
  OBJ=build/$(basename "$SRC" .cpp).o
  objs="$objs $OBJ"
  stale=0
  for h in binding/*.h ../../include/pyg_hip.h; do [ "$h" -nt "$OBJ" ] && stale=1; done
  if [ ! -f "$OBJ" ] || [ "$SRC" -nt "$OBJ" ] || [ $stale = 1 ]; then
    echo "$CXX $SRC"
    $CXX -std=c++17 -O2 -fPIC -fvisibility=hidden -D_GLIBCXX_USE_CXX11_ABI=$ABI \
      -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 \
      -I../../include -Ibinding -I"$TORCH/include" -I"$TORCH/include/torch/csrc/api/include" -I/opt/rocm/include \
      -Wno-deprecated-declarations -c "$SRC" -o "$OBJ" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
$CXX -shared -fPIC $objs -o $OUT \
  -L"$TORCH/lib" -ltorch -ltorch_cpu -ltorch_hip -lc10 -lc10_hip \
  -ldl -L.. -lpyg_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$TORCH/lib"
echo "built $OUT"
