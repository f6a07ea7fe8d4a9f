"""Last N pyg_hip kernels of a rocprofv3 --kernel-trace CSV with start offsets: python tools/trace_tail.py <csv> [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'pyg_hip' in r['Kernel_Name']]
tail = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 40):]
t0 = int(tail[0]['Start_Timestamp'])
for r in tail:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('pyg_hip::', '').replace('(anonymous namespace)::', '')[:56]
    print(f"{(st - t0) / 1000:8.1f} us {(en - st) / 1000:7.1f} us q={r.get('Queue_Id', '?')} blocks={int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1):6d} {name}")
