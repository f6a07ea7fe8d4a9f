import csv, glob, sys, json
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(list)); dur=defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(d+'/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r['Kernel_Name']].append(float(r['End_Timestamp'])-float(r['Start_Timestamp']))
for k,cs in acc.items():
    if not any(t in k for t in (sys.argv[0:0] or ['wide256', 'ticket', 'rgcn_fused', 'rgcn_grouped', 'regw'])): continue
    e={c: sum(v)/len(v) for c,v in cs.items()}
    e['avg_us']=sum(dur[k])/len(dur[k])/1e3
    print(k[:80]); print(json.dumps(e, indent=1))
