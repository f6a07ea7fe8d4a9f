#!/bin/bash
# batched samplers against the number of lanes (PYG_HIP_SAMPLER_LANES)
R=/root/repo/gpurun_out/r6_lanes
mkdir -p $R
cd /root/repo
for l in 1 2 3 4 6 8; do
  PYG_HIP_SAMPLER_LANES=$l python tools/bench_sampler_batched.py 16 2>&1 | grep -v amdgpu.ids | tee -a $R/c3.txt
  PYG_HIP_SAMPLER_LANES=$l python tools/bench_hetero_batched.py 8 2>&1 | grep -v amdgpu.ids | tee -a $R/c5.txt
done
