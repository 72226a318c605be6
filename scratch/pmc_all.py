import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = (r['Kernel_Name'].replace('(anonymous namespace)::', '')[:34], r.get('Grid_Size', r.get('Grid_Size_X', '?')))
    agg[k][r['Counter_Name']] += float(r['Counter_Value']); disp[k].add(r['Dispatch_Id'])
for k, v in sorted(agg.items()):
    if 'at::' in k[0] or 'rocclr' in k[0]: continue
    print('%-36s grid=%-9s n=%4d ' % (k[0], k[1], len(disp[k])) + '  '.join('%s/launch=%.1f KB' % (c, x / len(disp[k])) for c, x in sorted(v.items())))
