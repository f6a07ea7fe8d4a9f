#!/bin/bash
# kernel timeline of one C3 batch (rocprofv3 --kernel-trace) + full pytest -m gpu pass as the round's base line
R=/root/repo/gpurun_out/r6_c
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/trace -o t -- python /root/repo/tools/sampler_quick.py 12 > $R/trace.log 2>&1
python /root/repo/tools/trace_batch.py $(find $R/trace -name "*kernel_trace.csv" | head -1) > $R/c3_timeline.txt 2>&1
rm -rf $R/trace
cat $R/c3_timeline.txt
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > $R/pytest_full.txt 2>&1
echo "pytest rc=$?" >> $R/pytest_full.txt
tail -4 $R/pytest_full.txt
