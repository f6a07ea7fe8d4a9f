#!/bin/bash
# same-box A/B: gather_csr of narrow rows with one lane per item (g0) / 8 lanes from 8 positions per row (g8) / from 3 (g3)
cd /root/repo
cp pyg_lib_amd/libpyg_hip.so pyg_lib_amd/libpyg_hip_g8.so
for v in g0 g8 g3; do cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so; echo "== $v"; timeout 300 python tools/narrow_row_kernels.py 2>&1 | grep -v amdgpu | sed 's/sum.*| gather/gather/'; done
cp pyg_lib_amd/libpyg_hip_g8.so pyg_lib_amd/libpyg_hip.so
