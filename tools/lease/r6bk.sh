#!/bin/bash
# candidate blocks dealt densely: tests, stress, probes, kernel times
R=/root/repo/gpurun_out/r6_bk
mkdir -p $R
cd /root/repo
timeout 900 python -m pytest tests/test_rgcn_grouped_gpu.py tests/test_rgcn_gpu.py tests/test_rgcn_csc_gpu.py -m gpu -x -q > $R/pytest.txt 2>&1
echo "pytest rc=$?" >> $R/pytest.txt
tail -3 $R/pytest.txt
timeout 600 python tools/stress_rgcn_grouped.py > $R/stress.txt 2>&1; tail -1 $R/stress.txt
cd /tmp && export TMPDIR=/tmp
for a in "128" "256" "128 f32"; do
  PYTHONPATH=/root/repo rocprofv3 --kernel-trace --stats --output-format csv -d $R/p -o s -- python /root/repo/tools/rgcn_grouped_probe.py 30 15,10 $a > "$R/probe_${a// /_}.txt" 2>&1
  f=$(find $R/p -name "*kernel_stats.csv" | head -1)
  echo "F=$a: $(grep 'grouped=True' "$R/probe_${a// /_}.txt") $(python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'rgcn_grouped' in r['Name'] or 'rgcn_rowstart' in r['Name']:
        print('%s %.1f us x %s |' % (r['Name'].split('::')[-1][:28], float(r['AverageNs']) / 1e3, r['Calls']), end=' ')
PY
)" | tee -a $R/kern.txt
  rm -rf $R/p
done
