#!/bin/bash
# general (K, M) of the atomic-free layer: tests, stress, probes
R=/root/repo/gpurun_out/r6_bq
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_rgcn_grouped_gpu.py tests/test_rgcn_gpu.py tests/test_rgcn_csc_gpu.py tests/test_capi_raw_gpu.py tests/test_deterministic_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -3 $R/pytest.txt
timeout 900 python tools/stress_rgcn_grouped.py 600 2 > $R/stress.txt 2>&1; tail -3 $R/stress.txt
for a in "128" "256" "128 f32"; do
PYTHONPATH=/root/repo timeout 300 python tools/rgcn_grouped_probe.py 50 15,10 $a > "$R/probe_${a// /_}.txt" 2>&1
grep "grouped=True" "$R/probe_${a// /_}.txt"
done
