#!/bin/bash
# ablations of the wide pipeline (PYG_HIP_RGCN_ABLATE bits: 1 no W loads, 2 no products, 4 no row loads, 8 no zero-row stores)
R=/root/repo/gpurun_out/r6_bg
mkdir -p $R
cd /root/repo
cp pyg_lib_amd/libpyg_hip.so /tmp/base.so
for v in base abl1 abl3 abl4 abl8 abl12 abl15; do
  if [ $v = base ]; then cp /tmp/base.so pyg_lib_amd/libpyg_hip.so; else cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so; fi
  for a in "128" "256" "128 f32"; do
    t=$(PYTHONPATH=/root/repo timeout 300 python tools/rgcn_grouped_probe.py 50 15,10 $a 2>&1 | grep "grouped=True")
    echo "$v F=$a: $t" | tee -a $R/abl.txt
  done
done
cp /tmp/base.so pyg_lib_amd/libpyg_hip.so
