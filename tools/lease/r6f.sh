#!/bin/bash
R=/root/repo/gpurun_out/r6_f
mkdir -p $R
cd /root/repo
python tools/fold_timing.py > $R/fold_timing.txt 2>&1
cat $R/fold_timing.txt | tail -9
