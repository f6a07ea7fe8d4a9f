"""Kernel timeline of a few C3 sampler batches from a rocprofv3 --kernel-trace csv:
    rocprofv3 --kernel-trace --output-format csv -d out -- python tools/sampler_quick.py 6
    python tools/sampler_trace.py out/*/*_kernel_trace.csv
Prints the kernels of the last batch with start offsets (us) and durations."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last occurrence of seed_insert_kernel marks the start of the last batch
starts = [i for i, r in enumerate(rows) if 'seed_insert_kernel' in r['Kernel_Name']]
b = starts[-2] if len(starts) > 1 else starts[-1]
e = starts[-1] if len(starts) > 1 else len(rows)
t0 = int(rows[b]['Start_Timestamp'])
for r in rows[b - 6:e]:
    n = r['Kernel_Name'].replace('pyg_hip::(anonymous namespace)::', '').replace('pyg_hip::', '')
    n = n.split('(')[0][:60]
    st = (int(r['Start_Timestamp']) - t0) / 1e3
    du = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f'{st:9.1f} +{du:7.1f}  q={r.get("Queue_Id", "?"):>3s}  {n}')
