#!/bin/bash
# same-box A/B: softmax_csr's LDS-streamed kernel from 12 positions per group (ss12) / from 33 (ss33: shorter groups in registers)
cd /root/repo
cp pyg_lib_amd/libpyg_hip.so pyg_lib_amd/libpyg_hip_ss12.so
for v in ss12 ss33; do cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so; echo "== $v"; timeout 300 python tools/narrow_softmax_kernels.py 2>&1 | grep -v amdgpu; done
cp pyg_lib_amd/libpyg_hip_ss12.so pyg_lib_amd/libpyg_hip.so
timeout 400 python -m pytest tests/test_csr_gpu.py tests/test_csr_fuzz_gpu.py -m gpu -q 2>&1 | tail -1
