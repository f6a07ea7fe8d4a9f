#!/bin/bash
R=/root/repo/gpurun_out/r6_g
mkdir -p $R
cd /root/repo
for v in 0 1; do
  echo "HIP_FORCE_DEV_KERNARG=$v" >> $R/kernarg.txt
  HIP_FORCE_DEV_KERNARG=$v python tools/sampler_quick.py 60 2>&1 | tail -1 | cut -c1-420 >> $R/kernarg.txt
done
echo "default" >> $R/kernarg.txt
python tools/sampler_quick.py 60 2>&1 | tail -1 | cut -c1-420 >> $R/kernarg.txt
cat $R/kernarg.txt
cd /tmp && export TMPDIR=/tmp
HIP_FORCE_DEV_KERNARG=1 rocprofv3 --kernel-trace --output-format csv -d $R/trace -o t -- python /root/repo/tools/sampler_quick.py 12 > $R/trace.log 2>&1
python /root/repo/tools/trace_batch.py $(find $R/trace -name "*kernel_trace.csv" | head -1) > $R/c3_timeline_devkernarg.txt 2>&1
rm -rf $R/trace
cat $R/c3_timeline_devkernarg.txt
