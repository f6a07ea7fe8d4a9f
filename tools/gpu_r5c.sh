#!/bin/bash
R=/root/repo/gpurun_out/r5_c
mkdir -p $R
cd /root/repo
python tools/sampler_concurrency.py 40 > $R/conc_default.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/sampler_concurrency.py 40 > $R/conc_q8.txt 2>&1
GPU_MAX_HW_QUEUES=16 python tools/sampler_concurrency.py 40 > $R/conc_q16.txt 2>&1
cat $R/conc_default.txt $R/conc_q8.txt $R/conc_q16.txt | grep -v amdgpu.ids
