#!/bin/bash
# same-box A/B of the CSR row kernels: loads sunk to their uses (nopin) / pin_all with 4 positions per trip / with 8
R=/root/repo/gpurun_out/r6_pin
mkdir -p $R
cd /root/repo
cp pyg_lib_amd/libpyg_hip.so pyg_lib_amd/libpyg_hip_pin4.so
for round in 1 2; do
  for v in nopin pin4 u8; do
    cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so
    echo "== $v (round $round)" >> $R/scatter_time.txt
    timeout 200 python tools/scatter_time.py >> $R/scatter_time.txt 2>&1
  done
done
for v in nopin pin4 u8; do
  cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so
  timeout 300 python tools/csr_shape_sweep.py > $R/csr_shape_sweep_$v.txt 2>&1
  timeout 100 python tools/hub_sweep.py > $R/hub_sweep_$v.txt 2>&1
done
cp pyg_lib_amd/libpyg_hip_pin4.so pyg_lib_amd/libpyg_hip.so
timeout 600 python -m pytest tests/test_csr_gpu.py tests/test_reduce_gpu.py tests/test_capi_raw_gpu.py tests/test_deterministic_gpu.py -m gpu -q > $R/pytest.txt 2>&1
tail -1 $R/pytest.txt
grep -v amdgpu.ids $R/scatter_time.txt
