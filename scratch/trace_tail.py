import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# split into calls at corner_score
calls = []
for r in rows:
    if 'corner_score' in r['Kernel_Name']: calls.append([])
    if calls: calls[-1].append(r)
for idx in (5,):
    if idx >= len(calls): continue
    c = calls[idx]; t0 = int(c[0]['Start_Timestamp'])
    print('--- call', idx, 'total', (int(c[-1]['End_Timestamp']) - t0) / 1e3, 'us')
    for r in c:
        print('%9.1f %8.1f  %s grid=%s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:40], r.get('Grid_Size_X','')))
