"""Per-kernel averages of rocprofv3 --pmc passes:  python tools/pmc_summary.py <dir_FETCH> <dir_WRITE> [out.json]
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled for 16 B/lane coalesced streaming reads on
gfx950 (MI355X_MICROARCH.md, HBM section) -- reported raw AND corrected, since gathers are not that pattern."""
import csv, glob, json, sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    return acc


def short(n):
    n = n.replace('pyg_hip::(anonymous namespace)::', '').replace('pyg_hip::', '')
    return n.split('(')[0][:70]


res = {}
for d in sys.argv[1:3]:
    for k, cs in load(d).items():
        if 'pyg_hip' not in k:
            continue
        e = res.setdefault(short(k), {})
        for c, v in cs.items():
            e[c + '_KiB_avg'] = round(sum(v) / len(v), 1)
            e['launches'] = len(v)
for k, e in res.items():
    f, w = e.get('FETCH_SIZE_KiB_avg'), e.get('WRITE_SIZE_KiB_avg')
    if f is not None and w is not None:
        e['hbm_MB_raw'] = round((f + w) * 1024 / 1e6, 2)
        e['hbm_MB_fetch_x2'] = round((2 * f + w) * 1024 / 1e6, 2)
out = json.dumps(res, indent=1)
print(out)
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write(out)
