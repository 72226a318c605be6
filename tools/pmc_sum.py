"""Summarise a `rocprofv3 --pmc ...` output directory: one line per (kernel, grid size) with every counter averaged per
dispatch.  usage: python3 tools/pmc_sum.py <dir> [kernel-substring]"""
import collections
import csv
import glob
import re
import sys

files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not files:
    sys.exit("no *counter_collection.csv under " + sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        name = re.sub(r"\(.*", "", name)
        if want not in name:
            continue
        key = (name, int(r.get("Grid_Size", 0) or 0))
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[key].add(r["Dispatch_Id"])
for key in sorted(agg):
    n = len(disp[key])
    print("%-40s grid=%-9d n=%-4d " % (key[0][:40], key[1], n) +
          "  ".join("%s/launch=%.1f" % (c, v / n) for c, v in sorted(agg[key].items())))
