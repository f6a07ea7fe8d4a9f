#!/bin/bash
# round 5: one full GPU pass (one lease per call)
R=/root/repo/gpurun_out/r5_o$1
mkdir -p $R
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
cp gpurun_out/gpu_health.txt $R/gpu_health.txt 2>/dev/null
tail -5 $R/pytest.txt | grep -v "^$"
