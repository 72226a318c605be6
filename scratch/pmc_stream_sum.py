"""Per-kernel averages of a rocprofv3 --pmc run of scratch/stream_scale.py (one line per kernel and counter)."""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    if 'rocclr' in r['Kernel_Name'] or 'at::' in r['Kernel_Name']: continue
    k = (r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0], r['Grid_Size'])
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
for k, v in agg.items():
    print('%-26s grid=%-10s n=%2d  ' % (k[0][:26], k[1], len(disp[k])) + '  '.join('%s=%.1f' % (c, x / len(disp[k])) for c, x in sorted(v.items())))
