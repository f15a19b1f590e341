"""Per-kernel averages of rocprofv3 --pmc counters (counter_collection.csv), keyed by kernel name and grid."""
import csv, collections, glob, sys
path = sys.argv[1]
f = (glob.glob(path + '/*/*_counter_collection.csv') + glob.glob(path + '/*_counter_collection.csv'))[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
  n = r['Kernel_Name'].replace('void jpdse::', '').replace('jpdse::', '').split('(')[0][:56]
  key = (n, r['Grid_Size'])
  d = agg.setdefault(key, {})
  c = d.setdefault(r['Counter_Name'], [0, 0.0])
  c[0] += 1
  c[1] += float(r['Counter_Value'])
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for (n, g), d in agg.items():
  if flt and flt not in n:
    continue
  print('%-58s grid %9s  ' % (n, g) + '  '.join('%s avg %.1f (x%d)' % (k, v[1] / v[0], v[0]) for k, v in d.items()))
