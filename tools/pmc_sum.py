import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:50]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
disp = set()
for r in csv.DictReader(open(f)):
    key = (r['Kernel_Name'][:50], r['Dispatch_Id'])
    if key not in disp: disp.add(key); cnt[r['Kernel_Name'][:50]] += 1
for k, v in agg.items():
    if sys.argv[2] in k:
        print(k, 'dispatches', cnt[k])
        for c, x in sorted(v.items()): print('   %-28s %16.0f per dispatch %14.1f' % (c, x, x / cnt[k]))
