#!/bin/bash
# same-box A/B: the LDS-streamed CSR kernel for rows of any average length (ma0) / from 6 positions (ma6) / from 12 (ma12)
cd /root/repo
cp pyg_lib_amd/libpyg_hip.so pyg_lib_amd/libpyg_hip_ma0.so
for v in ma0 ma6 ma12; do cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so; echo "== $v"; timeout 250 python tools/narrow_row_kernels.py 2>&1 | grep -v amdgpu | grep "K=  1\|K=  2\|K=  3\|K=  5\|bfloat16  K=  4"; done
cp pyg_lib_amd/libpyg_hip_ma0.so pyg_lib_amd/libpyg_hip.so
