#!/bin/bash
R=/root/repo/gpurun_out/r6_bo
mkdir -p $R
cd /root/repo
cp pyg_lib_amd/libpyg_hip.so /tmp/base.so
for v in t16af; do
  cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so
  echo "== $v" | tee -a $R/wg.txt
  PYTHONPATH=/root/repo timeout 300 python tools/rgcn_wg_times.py 128 2>&1 | grep -v amdgpu.ids | tee -a $R/wg.txt
done
cp /tmp/base.so pyg_lib_amd/libpyg_hip.so
