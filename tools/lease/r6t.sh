#!/bin/bash
R=/root/repo/gpurun_out/r6_t
mkdir -p $R
cd /root/repo
PYG_HIP_SAMPLER_TRACE=1 python tools/trace_c5_host.py > $R/c5_host_trace.txt 2>&1
grep "trace us" $R/c5_host_trace.txt | tail -3
grep "sampler\] us:" $R/c5_host_trace.txt | tail -2
