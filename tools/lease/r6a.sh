#!/bin/bash
# round 6, first lease: the csc=True layer tests + the registry / grouped tests, then one bench line
R=/root/repo/gpurun_out/r6_b
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_rgcn_csc_gpu.py tests/test_rgcn_grouped_gpu.py tests/test_rgcn_gpu.py -m gpu -x -q > $R/pytest_rgcn.txt 2>&1
echo "pytest rc=$?" >> $R/pytest_rgcn.txt
tail -15 $R/pytest_rgcn.txt


