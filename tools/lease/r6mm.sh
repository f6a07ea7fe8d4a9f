#!/bin/bash
R=/root/repo/gpurun_out/r6_mm
mkdir -p $R
cd /root/repo
timeout 1500 python -m pytest tests/test_matmul_gen_gpu.py tests/test_matmul_fuzz_gpu.py tests/test_matmul_gpu.py tests/test_stress_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1
echo "rc=$?" >> $R/pytest.txt
tail -3 $R/pytest.txt
python tools/fuzz_matmul.py 2>&1 | tail -3
