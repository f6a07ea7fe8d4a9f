#!/bin/bash
R=/root/repo/gpurun_out/r5_e
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/trace -o t -- python /root/repo/tools/bench_sampler_batched.py 8 > $R/run.txt 2>&1
f=$(find $R/trace -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/trace_overlap.py $f 800 > $R/overlap.txt 2>&1
head -100 $R/overlap.txt
rm -rf $R/trace
