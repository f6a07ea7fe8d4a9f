#!/bin/bash
R=/root/repo/gpurun_out/r6_r
mkdir -p $R
cd /root/repo
for u in 4 8 12 4 8 12; do
  echo "U=$u $(PYG_HIP_CSR_U=$u python tools/scatter_time.py 2>/dev/null | tail -1)" | tee -a $R/csr_u.txt
done
timeout 600 python -m pytest tests/test_reduce_gpu.py tests/test_deterministic_gpu.py tests/test_csr_gpu.py -m gpu -x -q 2>&1 | tail -3
