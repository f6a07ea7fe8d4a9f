#!/bin/bash
R=/root/repo/gpurun_out/r5_d
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_sampler_batched_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -15 $R/pytest.txt
python tools/bench_sampler_batched.py > $R/b_default.txt 2>&1
PYG_HIP_SAMPLER_LANES=4 python tools/bench_sampler_batched.py 4 8 16 > $R/b_l4.txt 2>&1
PYG_HIP_SAMPLER_LANES=16 python tools/bench_sampler_batched.py 16 32 > $R/b_l16.txt 2>&1
GPU_MAX_HW_QUEUES=8 python tools/bench_sampler_batched.py 8 16 > $R/b_q8.txt 2>&1
GPU_MAX_HW_QUEUES=16 PYG_HIP_SAMPLER_LANES=16 python tools/bench_sampler_batched.py 8 16 32 > $R/b_q16.txt 2>&1
cat $R/b_*.txt | grep -v amdgpu.ids
