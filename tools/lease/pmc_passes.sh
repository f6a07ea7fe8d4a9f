#!/bin/bash
# SQ / TCC counter passes of one driver script (tools/pmc_*.py)
R=/root/repo/gpurun_out/r6_pmc2
mkdir -p $R
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/p1 -o p -- python /root/repo/tools/pmc_gen.py 4 > $R/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $R/p2 -o p -- python /root/repo/tools/pmc_gen.py 4 > $R/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/p3 -o p -- python /root/repo/tools/pmc_gen.py 4 > $R/p3.log 2>&1
python - $R <<'PY'
import csv, glob, sys, json
from collections import defaultdict
R = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for d in ('p1', 'p2', 'p3'):
    for f in glob.glob(f'{R}/{d}/**/*counter_collection.csv', recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if 'mfma_rows_gen' in r['Kernel_Name']]
        # dispatches in order: first 4 = K=100, next 4 = K=128
        ids = sorted({int(r['Dispatch_Id']) for r in rows})
        half = {i: ('K100' if n < len(ids) // 2 else 'K128') for n, i in enumerate(ids)}
        for r in rows:
            acc[half[int(r['Dispatch_Id'])]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}
print(json.dumps(out, indent=1))
open(f'{R}/gen_pmc.json', 'w').write(json.dumps(out, indent=1))
PY
tail -3 $R/p1.log
rm -rf $R/p1 $R/p2 $R/p3
