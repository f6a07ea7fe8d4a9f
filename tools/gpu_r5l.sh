#!/bin/bash
R=/root/repo/gpurun_out/r5_l
mkdir -p $R
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -7 $R/pytest.txt
