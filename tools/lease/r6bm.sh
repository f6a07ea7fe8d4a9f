#!/bin/bash
R=/root/repo/gpurun_out/r6_bm
mkdir -p $R
cd /root/repo
cp pyg_lib_amd/libpyg_hip.so /tmp/base.so
for rep in 1 2; do
for v in base nodense head; do
  if [ $v = base ]; then cp /tmp/base.so pyg_lib_amd/libpyg_hip.so; else cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so; fi
  t=$(PYTHONPATH=/root/repo timeout 300 python tools/rgcn_grouped_probe.py 50 15,10 128 2>&1 | grep "grouped=True")
  echo "$v: $t" | tee -a $R/ab.txt
done
done
cp /tmp/base.so pyg_lib_amd/libpyg_hip.so
