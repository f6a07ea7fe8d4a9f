#!/bin/bash
# kernel times (rocprofv3) of the ablated layer kernels
R=/root/repo/gpurun_out/r6_bh
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
cp /root/repo/pyg_lib_amd/libpyg_hip.so /tmp/base.so
for v in base abl15 abl12 abl4 abl3; do
  if [ $v = base ]; then cp /tmp/base.so /root/repo/pyg_lib_amd/libpyg_hip.so; else cp /root/repo/pyg_lib_amd/libpyg_hip_$v.so /root/repo/pyg_lib_amd/libpyg_hip.so; fi
  for a in "128" "256"; do
    PYTHONPATH=/root/repo rocprofv3 --kernel-trace --stats --output-format csv -d $R/p -o s -- python /root/repo/tools/rgcn_grouped_probe.py 20 15,10 $a > $R/log.txt 2>&1
    f=$(find $R/p -name "*kernel_stats.csv" | head -1)
    echo "$v F=$a: $(python - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'rgcn_grouped' in r['Name'] or 'rgcn_rowstart' in r['Name']:
        print('%s %.1f us x %s |' % (r['Name'].split('::')[-1][:28], float(r['AverageNs']) / 1e3, r['Calls']), end=' ')
PY
)" | tee -a $R/kern.txt
    rm -rf $R/p
  done
done
cp /tmp/base.so /root/repo/pyg_lib_amd/libpyg_hip.so
