#!/bin/bash
# forward + backward of the C5 layer under rocprofv3 --stats (tools/c5_train_step.py), default and atomic-free dX
R=/root/repo/gpurun_out/r6_p
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
for m in grouped atomic; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/prof_$m -o s -- python /root/repo/tools/c5_train_step.py 20 $m > $R/step_$m.log 2>&1
tail -1 $R/step_$m.log
f=$(find $R/prof_$m -name "*kernel_stats.csv" | head -1)
python - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:16]:
    print('%8.1f us/call x %5s  %s' % (float(r['AverageNs']) / 1e3, r['Calls'], r['Name'][:90]))
PY
cp $f $R/kernel_stats_$m.csv; rm -rf $R/prof_$m
done
