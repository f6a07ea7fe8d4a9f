#!/bin/bash
R=/root/repo/gpurun_out/r6_m
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
for t in 1 0; do
export PYG_HIP_SAMPLER_TERMINAL=$t
rocprofv3 --kernel-trace --output-format csv -d $R/trace -o t -- python /root/repo/tools/sampler_quick.py 12 > $R/trace.log 2>&1
python /root/repo/tools/trace_batch.py $(find $R/trace -name "*kernel_trace.csv" | head -1) > $R/c3_timeline_$t.txt 2>&1
rm -rf $R/trace
echo "terminal=$t"; grep "terminal\|scan_kernel<0\|sample_kernel<8" $R/c3_timeline_$t.txt
python /root/repo/tools/sampler_quick.py 60 2>&1 | tail -1 | cut -c1-120
done
