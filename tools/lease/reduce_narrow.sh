#!/bin/bash
R=/root/repo/gpurun_out/r6_narrow
mkdir -p $R
cd /root/repo
timeout 1200 python -m pytest tests/test_reduce_gpu.py tests/test_deterministic_gpu.py tests/test_stress_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1; tail -2 $R/pytest.txt
python tools/reduce_shape_sweep.py 2>&1 | grep -v amdgpu | tee $R/sweep.txt
