#!/bin/bash
R=/root/repo/gpurun_out/r6_ad
mkdir -p $R
cd /root/repo
for i in 1 2; do python tools/gen_time.py 2>/dev/null | tail -1 | tee -a $R/gen_time.txt; done
timeout 900 python -m pytest tests/test_matmul_gen_gpu.py tests/test_matmul_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -3
