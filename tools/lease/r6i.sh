#!/bin/bash
R=/root/repo/gpurun_out/r6_i
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
for mode in warm onepass0 plain; do
  unset PYG_HIP_SAMPLER_WARM PYG_HIP_SAMPLER_ONEPASS
  [ $mode = warm ] && export PYG_HIP_SAMPLER_WARM=1
  [ $mode = onepass0 ] && export PYG_HIP_SAMPLER_ONEPASS=0
  rocprofv3 --kernel-trace --output-format csv -d $R/trace_$mode -o t -- python /root/repo/tools/sampler_quick.py 12 > $R/trace_$mode.log 2>&1
  python - $(find $R/trace_$mode -name "*kernel_trace.csv" | head -1) > $R/timeline_$mode.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'pyg_hip' in r['Kernel_Name']]
tail = rows[-40:]
t0 = int(tail[0]['Start_Timestamp'])
for r in tail:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('pyg_hip::', '').replace('(anonymous namespace)::', '')[:48]
    print(f"{(st - t0) / 1000:8.1f} us {(en - st) / 1000:7.1f} us q={r.get('Queue_Id','?')} blocks={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d} {name}")
PY
  rm -rf $R/trace_$mode
  echo "== $mode"; tail -26 $R/timeline_$mode.txt
done
