#!/bin/bash
R=/root/repo/gpurun_out/r5_g
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_matmul_gpu.py tests/test_stress_gpu.py tests/test_sampler_gpu.py tests/test_rgcn_gpu.py -m gpu -x -q -k "c4 or k256 or 256 or grouped or dist or default_mode or full_size_c5 or checked" > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -4 $R/pytest.txt
for i in 1 2 3 4 5 6; do
  echo "stagger=1: $(python tools/c4_time.py 2>/dev/null | tail -1)" >> $R/c4.txt
  echo "stagger=0: $(PYG_HIP_MM_STAGGER=0 python tools/c4_time.py 2>/dev/null | tail -1)" >> $R/c4.txt
done
cat $R/c4.txt
