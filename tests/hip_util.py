"""Shared helpers for the GPU parity tests (tests/ may use oracle/; the product may not)."""
import numpy as np
import torch

import jpdse_hip
from jpdse_hip import ops, F32, BF16
from jpdse_hip.ops import Act

DEV = torch.device('cuda', 0)
DTYPES = [F32, BF16]
# Bounds of the per-operator comparisons (one kernel call against torch-CPU on the same rounded operands), max-norm
# (max|a-b| / max|b|) and element-wise (|a-b| <= tol (rms(b) + |b|)).  north_star asks for 1e-3 relative in fp32, forward AND
# backward; the bounds below are ~10x what the hardware measured over every case of the GPU suite
# (profiles/r03_parity_report.txt), so a dropped low-order term or a wrong tap weight of relative size 1e-4 fails:
#   fp32: worst max-norm 5.5e-6, worst element 5.9e-6 (conv fwd / dgrad / wgrad / bias grad, InstanceNorm, pools, losses)
#   bf16 (no reference counterpart; inputs, filters and outputs each rounded to 8 significant bits):
#         worst max-norm 7.0e-3 (fwd 3.7e-3 = one output rounding), worst element 1.5e-2 (head data gradient)
NORTH_STAR_F32 = 1e-3
RTOL = {F32: 5e-5, BF16: 1e-2}           # max-norm, forward and backward alike
ETOL = {F32: 6e-5, BF16: 2.5e-2}         # element-wise criterion of the same comparisons


def to_act(x_nchw, dtype):
  return ops.nchw_to_nhwc(x_nchw.to(DEV, torch.float32).contiguous(), dtype)


def to_nchw(act):
  return ops.nhwc_to_nchw(act).cpu()


def rel_err(a, b):
  a = torch.as_tensor(a, dtype=torch.float64)
  b = torch.as_tensor(b, dtype=torch.float64)
  return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


def elem_err(a, b):
  """Worst element of |a - b| / (rms(b) + |b|): the element-wise criterion |a - b| <= atol + rtol |b| with
  atol = tol * rms(b), rtol = tol, expressed as the smallest tol that would pass.  Unlike the max-norm figure of rel_err
  (one scalar per tensor, scaled by the LARGEST reference value) it sees an error confined to a low-magnitude region."""
  a = torch.as_tensor(a, dtype=torch.float64)
  b = torch.as_tensor(b, dtype=torch.float64)
  rms = b.pow(2).mean().sqrt().clamp_min(1e-30)
  return ((a - b).abs() / (rms + b.abs())).max().item()


# ---- parity report: every comparison made through this module, written at session end by tests/conftest.py
# (to $JPDSE_PARITY_REPORT, default gpurun_out/parity_report.txt; committed copies: profiles/rNN_parity_report.txt)
REPORT = []
CURRENT = ['']          # "<test function> <dtype>" of the running test (set by the autouse fixture in tests/conftest.py)
CASE = ['']             # its case id (the `case` / `name` parameter), stripped from the comparison's label and kept as a note


def record(what, measured, bound, note=''):
  w = str(what)
  if CASE[0] and w.startswith(CASE[0]):
    w = w[len(CASE[0]):].strip(' :')
  REPORT.append((CURRENT[0] + ' | ' + w, float(measured), float(bound), note or CASE[0]))


def _where_bad(a, b, tol):
  """Index ranges of the offending elements (diagnostic for intermittent failures)."""
  a = torch.as_tensor(a, dtype=torch.float64)
  b = torch.as_tensor(b, dtype=torch.float64)
  bad = torch.nonzero((a - b).abs() > tol * b.abs().max().clamp_min(1e-20))
  if bad.numel() == 0:
    return ''
  return ' [%d of %d elements off; index ranges %s; first %s]' % (
      bad.shape[0], a.numel(), [(int(bad[:, j].min()), int(bad[:, j].max())) for j in range(bad.shape[1])],
      bad[0].tolist())


def _etol_for(tol):
  """Element-wise bound that goes with a max-norm bound: the per-operator pairs (RTOL -> ETOL), otherwise the same number."""
  for dt in (F32, BF16):
    if abs(tol - RTOL[dt]) <= 1e-12 * RTOL[dt]:
      return ETOL[dt]
  return tol


def assert_close(a, b, tol, what='', elementwise=True, detail='', etol=None):
  """Two criteria: max|a-b| <= tol * max|b| (the max-norm figure the north star's "1e-3 rel" is read as) AND, element by
  element, |a-b| <= etol * (rms(b) + |b|) (etol: ETOL of the same dtype when tol is an RTOL entry, else tol).  `detail`
  (which window, which image) goes into the failure message and the report's note, not into the comparison's name."""
  e = rel_err(a, b)
  record(what + ' [max-norm]', e, tol, detail)
  assert e <= tol, '%s %s: max|a-b|/max|b| = %.3e > %.1e%s' % (what, detail, e, tol, _where_bad(a, b, tol))
  if elementwise:
    et = etol if etol is not None else _etol_for(tol)
    ee = elem_err(a, b)
    record(what + ' [element-wise]', ee, et, detail)
    assert ee <= et, '%s %s: worst element |a-b| / (rms(b) + |b|) = %.3e > %.1e' % (what, detail, ee, et)


def bf16_round(t):
  return t.to(torch.bfloat16).to(torch.float32)


def quantize_like(t, dtype):
  """What the device sees after storing `t` in the compute dtype."""
  return bf16_round(t) if dtype == BF16 else t


# ---- adjointness of a layer's three kernels at sizes the oracle cannot reach in seconds ----------------------------------
# <conv(x), dy> = <x, dgrad(dy)> = <w, wgrad(x, dy)>, residual quoted against the Cauchy-Schwarz scale |y| |dy|.
# The residual of a correct implementation is rounding noise (bf16 storage of y and dx; fp32 summation order of the weight
# gradient) whose size depends on the layer; ADJ_BOUND holds, per layer, 3x the largest residual measured over three seeds on
# MI355X (profiles/r03_parity_report.txt).  Layers without an entry use ADJ_DEFAULT.
ADJ_DEFAULT = 4e-6
ADJ_BOUND = {}
try:
  import json as _json, os as _os
  with open(_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'golden', 'adjointness_bounds.json')) as _fh:
    ADJ_BOUND = dict(_json.load(_fh)['bounds'])      # scripts/make_adj_bounds.py, from a parity report of the GPU suite
except (OSError, ValueError, KeyError):
  pass


def adjointness(name, N, H, W, C, K, k, st, pad, mode, seeds=(0, 1, 2), transposed=False, weight_scale=None):
  """Returns True when a second evaluation of the weight gradient was bit-identical (no atomics in any reduction)."""
  import zlib
  from jpdse_hip import ACT_NONE
  from jpdse_hip.layers import HipConv2d
  bound = ADJ_BOUND.get(name, ADJ_DEFAULT)
  repro = True
  for seed in seeds:
    g = torch.Generator(device=DEV).manual_seed(zlib.crc32(name.encode()) % 1000 + 7919 * seed)
    layer = HipConv2d(C, K, k, st, pad, mode, act=ACT_NONE, apply_bias=False, transposed=transposed, dtype=BF16, device=DEV)
    with torch.no_grad():
      ws = weight_scale if weight_scale is not None else 1.0 / (C * k * k) ** 0.5
      layer.weight.copy_(torch.randn(layer.weight.shape, generator=g, device=DEV) * ws)
    x = Act.empty(N, H, W, C, BF16, DEV)
    x.t.zero_()
    x.t[..., :C] = torch.randn((N, H, W, C), generator=g, device=DEV).to(torch.bfloat16)
    y, ctx = layer.fwd(x)
    Ky = y.C
    dy = y.empty_like()
    dy.t.zero_()
    dy.t[..., :Ky] = torch.randn(tuple(y.t.shape[:3]) + (Ky,), generator=g, device=DEV).to(torch.bfloat16)
    dx = layer.bwd(ctx, dy, need_dx=True, need_dw=True)
    torch.cuda.synchronize()
    dot = lambda a, b: (a.double() * b.double()).sum().item()
    wq = layer.weight.detach().to(torch.bfloat16)             # the packed panel holds the bf16-rounded master
    lhs, via_dx, via_dw = dot(y.t, dy.t), dot(x.t, dx.t), dot(wq, layer.weight.grad)
    scale = (dot(y.t, y.t) * dot(dy.t, dy.t)) ** 0.5          # Cauchy-Schwarz bound of the three products
    rx, rw = abs(lhs - via_dx) / scale, abs(lhs - via_dw) / scale
    record('adjointness <y,dy> vs <x,dx>: ' + name, rx, bound, 'seed %d; <y,dy>/scale %.1e' % (seed, lhs / scale))
    record('adjointness <y,dy> vs <w,dw>: ' + name, rw, bound, 'seed %d' % seed)
    assert scale > 0 and abs(lhs) < scale
    assert rx <= bound, '%s (seed %d): <y,dy> %.6e vs <x,dx> %.6e: %.2e of the scale > %.1e' % (name, seed, lhs, via_dx, rx, bound)
    assert rw <= bound, '%s (seed %d): <y,dy> %.6e vs <w,dw> %.6e: %.2e of the scale > %.1e' % (name, seed, lhs, via_dw, rw, bound)
    first = layer.weight.grad.detach().clone()
    layer.bwd(ctx, dy, need_dx=False, need_dw=True)
    torch.cuda.synchronize()
    repro = repro and torch.equal(first, layer.weight.grad)
    del layer, x, y, dy, dx, ctx, first
  return repro
