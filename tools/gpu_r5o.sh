#!/bin/bash
# round 5, step o: full GPU suite with the atomic-free fused layer in place
R=/root/repo/gpurun_out/r5_o
mkdir -p $R
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest_1.txt 2>&1
echo "pytest rc=$?" >> $R/pytest_1.txt
cp gpurun_out/gpu_health.txt $R/gpu_health_1.txt 2>/dev/null
tail -5 $R/pytest_1.txt | grep -v "^$"
