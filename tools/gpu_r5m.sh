#!/bin/bash
R=/root/repo/gpurun_out/r5_m
mkdir -p $R
cd /root/repo
timeout 600 python tools/stress_sampler_batched.py 400 > $R/stress_batched.txt 2>&1
tail -2 $R/stress_batched.txt
timeout 600 python tools/stress_sampler.py 2000 > $R/stress_single.txt 2>&1
tail -1 $R/stress_single.txt
for i in 1 2; do
  timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest_$i.txt 2>&1
  echo "pytest rc=$?" >> $R/pytest_$i.txt
  tail -4 $R/pytest_$i.txt | grep -v "^$"
done
