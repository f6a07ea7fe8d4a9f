#!/bin/bash
R=/root/repo/gpurun_out/r6_s
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/trace -o t -- python /root/repo/tools/prof_c5.py > $R/trace.log 2>&1
python /root/repo/tools/trace_tail.py $(find $R/trace -name "*kernel_trace.csv" | head -1) 26 > $R/c5_timeline.txt
rm -rf $R/trace
cat $R/c5_timeline.txt
