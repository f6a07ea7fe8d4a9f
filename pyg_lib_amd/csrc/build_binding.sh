#!/bin/bash
# Build libpyg.so: the torch operator library (namespace `pyg`) on top of the C-ABI libpyg_hip.so.
set -e
cd "$(dirname "$0")"
CXX=${CXX:-g++}
OUT=../libpyg.so
TORCH=$(python -c "import torch, os; print(os.path.dirname(torch.__file__))")
ABI=$(python -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
mkdir -p build
SRC=binding/pyg_binding.cpp
OBJ=build/pyg_binding.o
if [ ! -f "$OBJ" ] || [ "$SRC" -nt "$OBJ" ] || [ ../../include/pyg_hip.h -nt "$OBJ" ]; then
  echo "$CXX $SRC"
  $CXX -std=c++17 -O2 -fPIC -fvisibility=hidden -D_GLIBCXX_USE_CXX11_ABI=$ABI \
    -D__HIP_PLATFORM_AMD__=1 -DUSE_ROCM=1 \
    -I../../include -I"$TORCH/include" -I"$TORCH/include/torch/csrc/api/include" -I/opt/rocm/include \
    -Wno-deprecated-declarations -c "$SRC" -o "$OBJ"
fi
$CXX -shared -fPIC "$OBJ" -o $OUT \
  -L"$TORCH/lib" -ltorch -ltorch_cpu -ltorch_hip -lc10 -lc10_hip \
  -L.. -lpyg_hip -Wl,-rpath,'$ORIGIN' -Wl,-rpath,"$TORCH/lib"
echo "built $OUT"
