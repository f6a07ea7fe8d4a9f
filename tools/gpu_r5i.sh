#!/bin/bash
R=/root/repo/gpurun_out/r5_i
mkdir -p $R
cd /root/repo
(rocm-smi --showclocks --showpower --showperflevel 2>&1 | head -40) > $R/smi.txt
python tools/clock_probe.py > $R/clock.txt 2>&1
cat $R/clock.txt | grep -v amdgpu.ids
timeout 1200 python -m pytest tests -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -5 $R/pytest.txt
