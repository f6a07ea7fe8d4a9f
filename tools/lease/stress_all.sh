#!/bin/bash
# the long differential / stress tools in one lease (each under its own timeout)
R=/root/repo/gpurun_out/r6_stress
mkdir -p $R
cd /root/repo
run() { n=$1; shift; timeout 900 "$@" > $R/$n.txt 2>&1; echo "$n rc=$? $(tail -1 $R/$n.txt | cut -c1-200)" | tee -a $R/summary.txt; }
run fuzz_sampler python tools/fuzz_sampler.py 400 7
run fuzz_matmul python tools/fuzz_matmul.py 120 11
run stress_rgcn_grouped python tools/stress_rgcn_grouped.py 900 5
run stress_sampler python tools/stress_sampler.py 3000
run stress_sampler_batched python tools/stress_sampler_batched.py 150
run stress python tools/stress.py 300
run stress_atomics python tools/stress_atomics.py
run fuzz_reduce python tools/fuzz_reduce.py 500 1
for s in 1 2 3; do run stress_sampler_state_$s python tools/stress_sampler_state.py 150 $s; done
