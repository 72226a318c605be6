import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = [r for r in rows if 'rocclr' not in r['Kernel_Name'] and 'at::' not in r['Kernel_Name']]
def lv(h, w, n):
    out = [(h, w)]
    for _ in range(n): h, w = (h + 1) // 2, (w + 1) // 2; out.append((h, w))
    return out
cfg = {"kitti": (376, 1241, 3, 128, 21), "stress": (2160, 3840, 4, 2, 31)}
# kernels appear per build in a fixed order: pad_copy, pyr_down x levels, scharr, (borders), score
groups = collections.OrderedDict()
for r in rows:
    groups.setdefault((r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z']), []).append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in groups.items():
    v = sorted(v)
    print('%-34s grid=%-18s n=%2d  min %7.1f  med %7.1f us' % (k[0][:34], 'x'.join(k[1:]), len(v), v[0], v[len(v) // 2]))
for name, (h, w, n, B, win) in cfg.items():
    L = lv(h, w, n)
    print(name, 'B=%d' % B, 'algorithmic MB: pad_copy %.1f' % (2 * B * h * w / 1e6),
          ' '.join('down%d %.1f' % (i, B * (L[i][0] * L[i][1] + L[i + 1][0] * L[i + 1][1]) / 1e6) for i in range(n)),
          'scharr %.1f' % (5 * B * sum(a * b for a, b in L) / 1e6), 'score %.1f' % (5 * B * h * w / 1e6))
