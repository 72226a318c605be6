import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'rocclr' not in r['Kernel_Name'] and 'at::' not in r['Kernel_Name']]
H, W = 376, 1241
def lv(h, w, n):
    out = [(h, w)]
    for _ in range(n): h, w = (h + 1) // 2, (w + 1) // 2; out.append((h, w))
    return out
Ls = lv(H, W, 3)
groups = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    groups.setdefault((name, int(r['Grid_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z'])), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
seen = collections.defaultdict(int)
for (name, gx, gy, gz), v in groups.items():
    B = gz if name.startswith('corner_score') else gy
    if name == 'pad_copy_level0': mb = 2 * B * H * W / 1e6
    elif name == 'pyr_down_u8':
        i = seen[('d', B)]; seen[('d', B)] += 1
        mb = B * (Ls[i][0] * Ls[i][1] + Ls[i + 1][0] * Ls[i + 1][1]) / 1e6
    elif name == 'scharr_deriv_s16x2': mb = 5 * B * sum(a * b for a, b in Ls) / 1e6
    elif name.startswith('corner_score'): mb = 5 * B * H * W / 1e6
    else: mb = 0
    t = min(v)
    print('%-28s B=%5d  min %8.1f us  %8.1f MB algorithmic  %5.2f TB/s  (%4.1f %% of 8 TB/s)' % (name[:28], B, t, mb, mb / t, mb / t / 8 * 100) if mb else '%-28s B=%5d  min %8.1f us' % (name[:28], B, t))
