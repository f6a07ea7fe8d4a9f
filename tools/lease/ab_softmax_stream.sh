#!/bin/bash
# same-box A/B: softmax_csr through the LDS-streamed kernel for inner sizes below 64 bytes (sm64) / below 16 bytes (sm16) / never (sm0)
cd /root/repo
cp pyg_lib_amd/libpyg_hip.so pyg_lib_amd/libpyg_hip_sm64.so
for v in sm64 sm16 sm0; do cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so; echo "== $v"; timeout 250 python tools/narrow_softmax_kernels.py 2>&1 | grep -v amdgpu; done
cp pyg_lib_amd/libpyg_hip_sm64.so pyg_lib_amd/libpyg_hip.so
