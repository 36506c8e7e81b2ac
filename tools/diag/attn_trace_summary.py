import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'attn_fwd_cls' in r['Kernel_Name']]
d = [(int(rows[i]['End_Timestamp']) - int(rows[i]['Start_Timestamp'])) / 1e3 for i in idx]
print('attn_fwd_cls launches', len(d), 'min/median/max us', min(d), sorted(d)[len(d)//2], max(d))
# find a run of >= 20 consecutive attn launches (the back-to-back measurement) and print gaps
run = []
for a, b in zip(idx, idx[1:]):
    if b == a + 1: run.append(a)
print('consecutive pairs', len(run))
for a in run[:26]:
    r0, r1 = rows[a], rows[a + 1]
    print(round((int(r0['End_Timestamp']) - int(r0['Start_Timestamp'])) / 1e3, 1), 'us kernel, gap to next', round((int(r1['Start_Timestamp']) - int(r0['End_Timestamp'])) / 1e3, 1), 'us', 'grid', r0.get('Grid_Size'), 'wg', r0.get('Workgroup_Size'))
