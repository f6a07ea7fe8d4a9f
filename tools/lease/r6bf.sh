#!/bin/bash
R=/root/repo/gpurun_out/r6_bf
mkdir -p $R
cd /root/repo
timeout 600 python -m pytest tests/test_sampler_batched_gpu.py -m gpu -x -q > $R/pytest_batched.txt 2>&1; tail -2 $R/pytest_batched.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest_full.txt 2>&1
echo "pytest rc=$?" >> $R/pytest_full.txt
tail -3 $R/pytest_full.txt | grep -v "^$"
