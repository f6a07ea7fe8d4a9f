#!/bin/bash
# RNG stream carry: tests, timeline, timing
R=/root/repo/gpurun_out/r6_ab
mkdir -p $R
cd /root/repo
timeout 1200 python -m pytest tests/test_sampler_carry_gpu.py tests/test_sampler_gpu.py tests/test_sampler_fuzz_gpu.py tests/test_sampler_batched_gpu.py tests/test_biased_sampler_gpu.py tests/test_rgcn_gpu.py tests/test_stress_gpu.py tests/test_dist_helpers_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -12 $R/pytest.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/trace -o t -- python /root/repo/tools/sampler_quick.py 12 > $R/trace.log 2>&1
python /root/repo/tools/trace_batch.py $(find $R/trace -name "*kernel_trace.csv" | head -1) > $R/c3_timeline.txt 2>&1
rm -rf $R/trace
cat $R/c3_timeline.txt
cd /root/repo
python tools/sampler_quick.py 60 > $R/quick.json 2>&1; tail -c 900 $R/quick.json
python tools/trace_sampler_host.py 2> $R/host_trace.txt > /dev/null
tail -4 $R/host_trace.txt
