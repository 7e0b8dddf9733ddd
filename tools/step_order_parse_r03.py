"""Ordered kernel list of ONE replayed train step from a rocprofv3 --kernel-trace csv (+ the per-step aggregate).
usage: step_order_parse_r03.py <trace dir> <marker substring> [steps]   -> stdout"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
marker = sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
pos = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
per = collections.Counter()
print("marker", marker, "occurrences", len(pos))
a, b = pos[-steps - 1], pos[-1]
seg = rows[a:b]
t0, t1 = int(seg[0]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in seg)
print("per step: wall %.3f ms  sum of kernel durations %.3f ms  kernels %.0f" % ((t1 - t0) / 1e6 / steps, busy / 1e6 / steps, len(seg) / steps))
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    n = r['Kernel_Name']
    agg[n][0] += 1; agg[n][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print("top kernels per step (count, avg us, ms/step):")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print("  %6.1f %7.1f %7.3f  %s" % (c / steps, t / c / 1e3, t / 1e6 / steps, n[:130]))
one = rows[pos[-2]:pos[-1]]
print("ordered kernels of the last complete step (us, grid x block, name):")
for r in one:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    g = "%sx%sx%s/%s" % (r.get('Grid_Size_X', '?'), r.get('Grid_Size_Y', '?'), r.get('Grid_Size_Z', '?'), r.get('Workgroup_Size_X', '?'))
    print("  %8.1f  %-22s %s" % (d, g, r['Kernel_Name'][:120]))
