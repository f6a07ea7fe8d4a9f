#!/bin/bash
# same-box A/B: segment_*_csr of narrow rows of whole 16-byte slices with 8 lanes per item from 64 positions per row (vl64) / 16 / 8
cd /root/repo
cp pyg_lib_amd/libpyg_hip.so pyg_lib_amd/libpyg_hip_vl64.so
for v in vl64 vl16 vl8; do cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so; echo "== $v"; timeout 300 python tools/narrow_row_kernels.py 2>&1 | grep -v amdgpu | grep "float32   K=  4\|float32   K=  8\|float32   K= 12\|bfloat16  K=  8\|bfloat16  K= 16\|bfloat16  K= 24" | sed 's/| gather.*//'; done
cp pyg_lib_amd/libpyg_hip_vl64.so pyg_lib_amd/libpyg_hip.so
