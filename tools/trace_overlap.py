"""Do the kernels of concurrent sampler batches overlap?  python tools/trace_overlap.py <kernel_trace.csv> [last N rows]
Prints, for the tail of the trace: sum of kernel durations, the union of their intervals (device busy time), the span, the
number of distinct queues / streams seen, and a timeline excerpt."""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 600
rows = rows[-n:]
iv = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows]
total = sum(e - s for s, e in iv)
busy, cur_s, cur_e = 0, None, None
for s, e in sorted(iv):
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
span = max(e for s, e in iv) - min(s for s, e in iv)
keys = [k for k in ('Queue_Id', 'Stream_Id') if k in rows[0]]
print(f'{len(rows)} kernels: sum {total / 1e3:.1f} us, union {busy / 1e3:.1f} us, span {span / 1e3:.1f} us, '
      f'overlap factor {total / busy:.2f}; ' + ', '.join(f'{k}: {len({r[k] for r in rows})} distinct' for k in keys))
t0 = iv[0][0]
for r in rows[:90]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('pyg_hip::', '').replace('(anonymous namespace)::', '').replace('sampler::', '')[:44]
    q = '/'.join(r[k] for k in keys)
    print(f"{(st - t0) / 1000:9.1f} +{(en - st) / 1000:6.1f} us q={q:>8} blocks={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d} {name}")
