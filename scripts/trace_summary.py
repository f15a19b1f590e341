"""Summarise a rocprofv3 kernel trace: last full train step, grouped by kernel and grid."""
import csv, collections, sys, glob
path = sys.argv[1]
rows = list(csv.DictReader(open((glob.glob(path + '/*/*_kernel_trace.csv') + glob.glob(path + '/*_kernel_trace.csv'))[0])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
adam = [i for i, n in enumerate(names) if 'adam_kernel' in n]
lo, hi = adam[-3] + 1, adam[-1] + 1
step = rows[lo:hi]
dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(dur(r) for r in step)
span = int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])
print('kernels in step %d   busy %.2f ms   span %.2f ms' % (len(step), tot / 1e6, span / 1e6))
byk = collections.OrderedDict()
for r in step:
  n = r['Kernel_Name'].replace('void jpdse::', '').replace('jpdse::', '').split('(')[0][:58]
  a = byk.setdefault(n, [0, 0]); a[0] += 1; a[1] += dur(r)
print('--- by kernel')
for k, (c, d) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
  print('%-60s x%4d  %8.2f ms  %5.1f%%' % (k, c, d / 1e6, 100.0 * d / tot))
if len(sys.argv) > 3:
  agg = collections.OrderedDict()
  for r in step:
    n = r['Kernel_Name'].replace('void jpdse::', '').replace('jpdse::', '').split('(')[0][:66]
    key = (n, r['Grid_Size_X'], r['Grid_Size_Y'])
    a = agg.setdefault(key, [0, 0]); a[0] += 1; a[1] += dur(r)
  print('--- by kernel and grid')
  for k, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3])]:
    print('%-68s grid %8s x%3s  x%3d  %8.1f us  avg %7.1f' % (k[0], k[1], k[2], c, d / 1e3, d / 1e3 / c))
# gaps between consecutive kernels of the step (launch latency the GPU could not hide)
gaps = []
for p, q in zip(step[:-1], step[1:]):
  g = int(q['Start_Timestamp']) - int(p['End_Timestamp'])
  gaps.append((g, p['Kernel_Name'].replace('void jpdse::', '').split('(')[0][:40], q['Kernel_Name'].replace('void jpdse::', '').split('(')[0][:40]))
pos = [g for g in gaps if g[0] > 0]
print('--- gaps: %d of %d boundaries idle, %.2f ms in all; > 5 us: %d (%.2f ms); > 20 us: %d (%.2f ms)' % (
    len(pos), len(gaps), sum(g[0] for g in pos) / 1e6, sum(1 for g in pos if g[0] > 5000), sum(g[0] for g in pos if g[0] > 5000) / 1e6,
    sum(1 for g in pos if g[0] > 20000), sum(g[0] for g in pos if g[0] > 20000) / 1e6))
for g in sorted(pos, key=lambda t: -t[0])[:12]:
  print('   %7.1f us  after %-40s before %s' % (g[0] / 1e3, g[1], g[2]))
