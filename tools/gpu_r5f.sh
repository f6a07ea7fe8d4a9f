#!/bin/bash
R=/root/repo/gpurun_out/r5_f
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_sampler_gpu.py tests/test_sampler_batched_gpu.py tests/test_rgcn_gpu.py tests/test_biased_sampler_gpu.py tests/test_dist_helpers_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -4 $R/pytest.txt
python tools/bench_sampler_batched.py 8 16 32 > $R/b_default.txt 2>&1
PYG_HIP_SAMPLER_LANES=16 python tools/bench_sampler_batched.py 16 32 64 > $R/b_l16.txt 2>&1
cat $R/b_*.txt | grep -v amdgpu.ids
