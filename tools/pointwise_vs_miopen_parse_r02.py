import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
groups, cur, started = [], [], False
for r in rows:
    n = r['Kernel_Name']
    is_mark = 'FillFunctor' in n and int(r.get('Grid_Size', r.get('Grid_Size_X', 0)) or 0) > 0 and False
    cur.append(r)
# separators: fill kernels on the 12345-element tensor -> grid size unknown; use name + tiny duration + count heuristic
idx = [i for i, r in enumerate(rows) if 'FillFunctor<float>' in r['Kernel_Name']]
# the last 1 + 12*6 fills are the markers
idx = idx[-(1 + 12 * 6):]
shapes = [(160, 64, 64), (80, 128, 128), (80, 64, 64), (40, 256, 256), (40, 128, 128), (20, 512, 512),
          (20, 256, 256), (20, 512, 256), (20, 1024, 512), (40, 512, 256), (40, 256, 128), (80, 256, 128)]
labels = ['ours_wgrad', 'mi_wgrad', 'mi_dgrad', 'mi_fwd', 'ours_fwd', 'ours_dgrad']
print("%-18s " % "HxW K->N" + " ".join("%11s" % l for l in labels))
for si, sh in enumerate(shapes):
    out = []
    for gi in range(6):
        a, b = idx[si * 6 + gi], idx[si * 6 + gi + 1]
        seg = rows[a + 1:b]
        tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg) / 1e3 / 10
        out.append("%6.1f(%2d)" % (tot, len(seg) // 10) if seg else "     -    ")
    print("%3dx%-3d %4d->%-4d " % (sh[0], sh[0], sh[1], sh[2]) + " ".join("%11s" % o for o in out))
