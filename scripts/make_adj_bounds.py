"""Per-layer bounds of the adjointness tests from a parity report of `pytest -m gpu` (tests/conftest.py writes it):
bound = 3 x the largest residual measured for the layer over its seeds and both products (<x,dx>, <w,dw>), rounded up to two
significant digits, floor 5e-7 (a third of the smallest layer noise seen would make the bound a coin toss).
Usage: python scripts/make_adj_bounds.py gpurun_out/r3_parity_report.txt  ->  tests/golden/adjointness_bounds.json"""
import json, math, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
worst = {}
for line in open(src):
  m = re.match(r'^(\S+)( \S+)? \| adjointness <y,dy> vs <[xw],d[xw]>: (\S+)\s+(\d+)\s+([0-9.e+-]+)\s', line)
  if m:
    name, val = m.group(3), float(m.group(5))
    worst[name] = max(worst.get(name, 0.0), val)
def up2(x):
  e = math.floor(math.log10(x)) - 1
  return math.ceil(x / 10 ** e) * 10 ** e
bounds = {k: float('%.2g' % max(up2(3.0 * v), 5e-7)) for k, v in sorted(worst.items())}
out = os.path.join(ROOT, 'tests', 'golden', 'adjointness_bounds.json')
json.dump(dict(source=os.path.basename(src), rule='3 x max residual over seeds and products, floor 5e-7', measured=worst, bounds=bounds),
          open(out, 'w'), indent=1, sort_keys=True)
for k in sorted(bounds):
  print('%-28s measured %.3e -> bound %.1e' % (k, worst[k], bounds[k]))
