#!/bin/bash
# SQ counter passes of one driver script:  tools/pmc_sq.sh <out dir under gpurun_out> <kernel name substring> <python script> [args]
# (three separate --pmc passes + kernel trace; averages per launch of the kernels whose name contains the substring)
R=/root/repo/gpurun_out/$1; K=$2; shift 2
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/p1 -o p -- python "$@" > $R/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/p2 -o p -- python "$@" > $R/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/p3 -o p -- python "$@" > $R/p3.log 2>&1
python - $R "$K" <<'PY'
import csv, glob, sys, json
from collections import defaultdict
R, K = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(list)); dur = defaultdict(list)
for d in ('p1', 'p2', 'p3'):
    for f in glob.glob(f'{R}/{d}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if K in r['Kernel_Name']:
                acc[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(f'{R}/{d}/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if K in r['Kernel_Name']:
                dur[r['Kernel_Name'][:90]].append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
out = {}
for k, cs in acc.items():
    e = {c: sum(v) / len(v) for c, v in cs.items()}
    e['avg_us'] = sum(dur[k]) / max(len(dur[k]), 1) / 1e3
    e['launches'] = len(dur[k]) // 3
    out[k] = e
print(json.dumps(out, indent=1))
open(f'{R}/sq.json', 'w').write(json.dumps(out, indent=1))
PY
rm -rf $R/p1 $R/p2 $R/p3
