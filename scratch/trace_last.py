import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'rocclr' not in r['Kernel_Name'] and 'at::' not in r['Kernel_Name']]
last = rows[-int(sys.argv[2]):]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    print('%9.1f %8.1f  %s grid=%s' % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'].replace('(anonymous namespace)::','')[:40], r.get('Grid_Size_X', '')))
