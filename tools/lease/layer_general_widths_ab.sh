#!/bin/bash
# the run-time-size instance through the sub-item pipeline: tests, then A/B against the item-at-a-time walk (libpyg_hip_head.so)
R=/root/repo/gpurun_out/r6_br
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_rgcn_grouped_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -3 $R/pytest.txt
timeout 900 python tools/stress_rgcn_grouped.py 400 3 > $R/stress.txt 2>&1; tail -1 $R/stress.txt
cp pyg_lib_amd/libpyg_hip.so /tmp/base.so
for v in base head base head; do
  if [ $v = base ]; then cp /tmp/base.so pyg_lib_amd/libpyg_hip.so; else cp pyg_lib_amd/libpyg_hip_$v.so pyg_lib_amd/libpyg_hip.so; fi
  for a in 64 32 192; do
    t=$(PYTHONPATH=/root/repo timeout 300 python tools/rgcn_grouped_probe.py 50 15,10 $a 2>&1 | grep "grouped=")
    echo "$v F=$a: $t" | tr '\n' ' ' | tee -a $R/ab.txt; echo | tee -a $R/ab.txt
  done
done
cp /tmp/base.so pyg_lib_amd/libpyg_hip.so
