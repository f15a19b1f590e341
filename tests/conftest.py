import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'jpd-se_amd')
for p in (ROOT, PKG):
  if p not in sys.path:
    sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
  # The CPU oracle runs small torch graphs; torch's default thread count is the host's core count (128 on the GPU boxes, of
  # which a one-GPU job may use 16): oversubscribed, a 64x128 oracle step took 2.2 s there against 0.15 s on 8 threads.
  try:
    import torch
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 8)
    torch.set_num_threads(max(1, min(16, n)))
  except Exception:
    pass


@pytest.fixture(scope='session')
def golden_dir():
  return GOLDEN


@pytest.fixture(autouse=True)
def _parity_report_context(request):
  """Tags the comparisons recorded by tests/hip_util.py with the running test function and its dtype parameter."""
  hu = sys.modules.get('hip_util')
  if hu is not None:
    params = getattr(getattr(request.node, 'callspec', None), 'params', {})
    dt = params.get('dtype')
    tag = {0: 'fp32', 1: 'bf16'}.get(dt, '') if isinstance(dt, int) else ''
    hu.CURRENT[0] = (request.node.originalname or request.node.name) + (' ' + tag if tag else '')
    case = params.get('case', params.get('name', ''))
    hu.CASE[0] = case[0] if isinstance(case, (tuple, list)) and case and isinstance(case[0], str) else (case if isinstance(case, str) else '')
  yield


def pytest_sessionfinish(session, exitstatus):
  """Parity report: what every hip_util comparison of this run measured next to its bound (tests/hip_util.py REPORT)."""
  hu = sys.modules.get('hip_util')
  rows = getattr(hu, 'REPORT', None)
  if not rows:
    return
  path = os.environ.get('JPDSE_PARITY_REPORT') or os.path.join(ROOT, 'gpurun_out', 'parity_report.txt')
  try:
    os.makedirs(os.path.dirname(path), exist_ok=True)
    agg = {}
    for what, measured, bound, note in rows:
      a = agg.setdefault(what, [0, -1.0, bound, note])
      a[0] += 1
      if measured / max(bound, 1e-300) > a[1] / max(a[2], 1e-300):    # keep the call that used most of its bound
        a[1], a[2], a[3] = measured, bound, note
    with open(path, 'w') as fh:
      fh.write('# parity report of one `pytest -m gpu` run: worst measured value over the calls of each comparison, next to its bound\n')
      fh.write('# %-148s %5s %11s %9s %7s  %s\n' % ('test | comparison', 'calls', 'measured', 'bound', 'used', 'worst case'))
      for what in sorted(agg):
        n, m, b, note = agg[what]
        fh.write('%-150s %5d %11.3e %9.1e %6.1f%%  %s\n' % (what[:150], n, m, b, 100.0 * m / b if b > 0 else 0.0, note))
  except OSError:
    pass
