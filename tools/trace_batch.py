"""Per-kernel timeline of the LAST sampler call in a rocprofv3 --kernel-trace CSV:  python tools/trace_batch.py <csv>"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'mt_prefix' in r['Kernel_Name']]  # the first launch of a call (side stream)
s, e = idx[-2], idx[-1]
t0 = int(rows[s]['Start_Timestamp'])
prev_end = None
for r in rows[s:e]:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('pyg_hip::', '').replace('(anonymous namespace)::', '')[:64]
    print(f"{(st - t0) / 1000:8.1f} us {(en - st) / 1000:7.1f} us  blocks={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d} vgpr={r['VGPR_Count']:>3} {name}")
