#!/bin/bash
R=/root/repo/gpurun_out/r5_h
mkdir -p $R
cd /root/repo
for i in 1 2 3 4 5 6 7 8 9 10; do
  echo "stagger=1 $(python tools/c4_time.py 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["rank0_launch"]["frac"])')" >> $R/c4.txt
  echo "stagger=0 $(PYG_HIP_MM_STAGGER=0 python tools/c4_time.py 2>/dev/null | tail -1 | python -c 'import sys,json; r=json.loads(sys.stdin.read()); print(r["rank0_launch"]["frac"])')" >> $R/c4.txt
done
sort $R/c4.txt | tr '\n' ' '
