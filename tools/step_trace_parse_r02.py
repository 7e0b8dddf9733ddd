import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
cnt = collections.Counter(r['Kernel_Name'] for r in rows)
# a kernel that runs once per step: pick our dense loss-gradient kernel if present
key = None
for n in cnt:
    if 'yolo_grad_dense' in n or 'yolo_dense' in n: key = n
if key is None:
    key = min((n for n in cnt if 30 <= cnt[n] <= 60), key=lambda n: cnt[n])
pos = [i for i, r in enumerate(rows) if r['Kernel_Name'] == key]
print("step marker:", key[:80], "occurrences", len(pos))
a, b = pos[-21], pos[-1]
seg = rows[a:b]; steps = 20
t0, t1 = int(seg[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
gaps = 0; hist = collections.Counter()
prev_end = int(seg[0]['End_Timestamp'])
for r in seg[1:]:
    s = int(r['Start_Timestamp'])
    g = s - prev_end
    if g > 0:
        gaps += g
        hist[min(int(g / 1000), 20)] += 1
    prev_end = max(prev_end, int(r['End_Timestamp']))
print("per step: wall %.3f ms  sum of kernel durations %.3f ms  idle gaps %.3f ms  kernels %.0f" % ((t1 - t0) / 1e6 / steps, busy / 1e6 / steps, gaps / 1e6 / steps, len(seg) / steps))
print("gap histogram (us -> count/step):", {k: round(v / steps, 1) for k, v in sorted(hist.items())})
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r['Kernel_Name']
    agg[n][0] += 1; agg[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print("top kernels per step (count, avg us, ms/step):")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 45]:
    print("  %6.1f %7.1f %7.3f  %s" % (c / steps, t / c / 1e3, t / 1e6 / steps, n[:110]))
