#!/bin/bash
# Builds oracle/_ref/libpyg_ref.so: the part of the REAL reference CPU path that compiles from its
# own sources with g++ and libtorch alone -- index_sort, scatter_*, segment_*_coo / gather_coo,
# segment_*_csr / gather_csr, softmax_csr (front + CPU kernels + autograd wrappers).  Sources are compiled where they lie under
# /root/reference; nothing is copied and no stand-in header is written.
#
# NOT built (unbuildable in this image, see DESIGN.md "Oracle"): ops/cpu/matmul_kernel.cpp and
# sampler/cpu/neighbor_kernel.cpp need the un-vendored parallel-hashmap submodule and the
# cmake-generated pyg_lib/csrc/config.h.
#
# The result is used (a) to generate tests/golden/*.npz (tests/golden/make_ref_golden.py) and
# (b) optionally as a "reference"-kind CPU baseline.  It never ships and is git-ignored.
set -e
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -d "$REF/pyg_lib/csrc" ] || { echo "no reference tree at $REF: skipping oracle/_ref"; exit 0; }
mkdir -p "$OUT/obj"
TORCH=$(python -c "import torch, os; print(os.path.dirname(torch.__file__))")
ABI=$(python -c "import torch; print(int(torch._C._GLIBCXX_USE_CXX11_ABI))")
SRCS="ops/index_sort ops/cpu/index_sort_kernel ops/scatter ops/cpu/scatter_kernel ops/autograd/scatter_kernel ops/segment_coo ops/cpu/segment_coo_kernel ops/autograd/segment_coo_kernel ops/segment_csr ops/cpu/segment_csr_kernel ops/autograd/segment_csr_kernel ops/softmax ops/cpu/softmax_kernel ops/autograd/softmax_kernel"
objs=""
pids=""
for s in $SRCS; do
  o="$OUT/obj/$(echo $s | tr / _).o"
  objs="$objs $o"
  if [ ! -f "$o" ] || [ "$REF/pyg_lib/csrc/$s.cpp" -nt "$o" ]; then
    g++ -std=c++20 -O2 -fPIC -fopenmp -D_GLIBCXX_USE_CXX11_ABI=$ABI -I"$REF" -I"$TORCH/include" \
      -I"$TORCH/include/torch/csrc/api/include" -Wno-deprecated-declarations \
      -c "$REF/pyg_lib/csrc/$s.cpp" -o "$o" &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
g++ -shared -fopenmp $objs -o "$OUT/libpyg_ref.so" -L"$TORCH/lib" -ltorch -ltorch_cpu -lc10 -Wl,-rpath,"$TORCH/lib"
echo "built $OUT/libpyg_ref.so"
