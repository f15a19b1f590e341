"""Merge the two rocprofv3 --pmc passes of scripts/run_pmc_hbm.sh (gpurun_out/pmc_hbm_FETCH_SIZE.txt, ..._WRITE_SIZE.txt:
per-dispatch averages in KB) into profiles/<tag>_hbm_traffic.txt and refresh profiles/hbm_traffic.json (the headline
kernel's bytes per launch, read by bench.py).  FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B:
MI355X_MICROARCH.md 'HBM'); WRITE_SIZE is exact.  Usage: python scripts/make_hbm_table.py r02"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r02'
pat = re.compile(r'^(.*?)\s+grid\s+(\d+)\s+(FETCH_SIZE|WRITE_SIZE) avg ([0-9.]+) \(x(\d+)\)')
tab = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
  for line in open(os.path.join(ROOT, 'gpurun_out', 'pmc_hbm_%s.txt' % c)):
    m = pat.match(line.rstrip())
    if m:
      key = (m.group(1).strip(), int(m.group(2)))
      tab.setdefault(key, {})[c] = (float(m.group(4)), int(m.group(5)))
rows = []
for (name, grid), d in tab.items():
  f, nf = d.get('FETCH_SIZE', (0.0, 0))
  w, nw = d.get('WRITE_SIZE', (0.0, 0))
  fetch_mb, write_mb = 2.0 * f * 1024 / 1e6, w * 1024 / 1e6
  rows.append((name, grid, max(nf, nw), fetch_mb, write_mb, (fetch_mb + write_mb) * max(nf, nw)))
rows.sort(key=lambda r: -r[5])
out = os.path.join(ROOT, 'profiles', '%s_hbm_traffic.txt' % tag)
with open(out, 'w') as fh:
  fh.write('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, scripts/run_pmc_hbm.sh) over `bench.py --steps 2 --warmup 1`,\n'
           'per-dispatch averages.  FETCH_SIZE (KB) is doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md \'HBM\'); WRITE_SIZE (KB)\n'
           'is exact.  These are memory-side (fabric) bytes of the L2s: Infinity-Cache hits are included.  Sorted by total bytes moved.\n\n')
  fh.write('%-62s %9s %6s %12s %12s %12s\n' % ('kernel', 'grid', 'calls', 'fetch MB', 'write MB', 'total MB'))
  for name, grid, n, f, w, _ in rows[:70]:
    fh.write('%-62s %9d %6d %12.1f %12.1f %12.1f\n' % (name[:62], grid, n, f, w, f + w))
head = [r for r in rows if r[0].startswith('gemm_halo_kernel<4, 2, 0, false') and r[1] == 131072]
if head:
  name, grid, n, f, w, _ = head[0]
  js = dict(kernel='gemm_halo_kernel<4,2> grid 131072 (ResnetBlock 3x3, N=1024 K=9216)', fetch_bytes_per_launch=f * 1e6,
            write_bytes_per_launch=w * 1e6, launches_averaged=n,
            source='profiles/%s_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE; FETCH_SIZE x2 per the gfx950 correction)' % tag)
  json.dump(js, open(os.path.join(ROOT, 'profiles', 'hbm_traffic.json'), 'w'), indent=1)
# norm family (InstanceNorm kernels of norm.hip) per step: the pass ran `--steps 2 --warmup 1` = 3 steps
norm = [r for r in rows if r[0].startswith(('inorm_', 'moment_kernel', 'finalize_')) ]
if norm:
  steps = 3.0
  nb = sum((f + w) * 1e6 * n for _, _, n, f, w, _ in norm) / steps
  tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
  js = json.load(open(tpath)) if os.path.exists(tpath) else {}
  js['norm_family_bytes_per_step'] = nb
  js['norm_family_kernels'] = sorted({r[0] for r in norm})
  json.dump(js, open(tpath, 'w'), indent=1)
  with open(out, 'a') as fh:
    fh.write('\nInstanceNorm family (inorm_*, moment_kernel, finalize_*): %.1f MB per step memory-side (3 steps in the pass)\n' % (nb / 1e6))
print(open(out).read()[:3000])
