#!/bin/bash
# round 5, step n: the atomic-free fused layer (rgcn_grouped.h): parity tests, the C5 layer with both kernels, kernel trace
R=/root/repo/gpurun_out/r5_n
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_rgcn_grouped_gpu.py tests/test_rgcn_gpu.py tests/test_deterministic_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
grep -v "^$" $R/pytest.txt | tail -25
export PYTHONPATH=/root/repo
timeout 300 python tools/rgcn_grouped_probe.py 50 > $R/probe.txt 2>&1
tail -4 $R/probe.txt
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/prof -o layer -- python /root/repo/tools/rgcn_grouped_probe.py 20 > $R/prof.txt 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('/root/repo/gpurun_out/r5_n/prof/**/*kernel_stats.csv', recursive=True):
    rows = list(csv.DictReader(open(f)))
    for r in rows[:12]:
        print(r['Name'][:90], r['Calls'], r['AverageNs'])
PY
