"""Shared helpers for the GPU parity tests (tests/ may use oracle/; the product may not)."""
import numpy as np
import torch

import jpdse_hip
from jpdse_hip import ops, F32, BF16
from jpdse_hip.ops import Act

DEV = torch.device('cuda', 0)
DTYPES = [F32, BF16]
# north_star: fp32 within 1e-3 relative of the torch-CPU oracle.  bf16 has no reference
# counterpart; its bound is the documented looser one below.
RTOL = {F32: 1e-3, BF16: 3e-2}


def to_act(x_nchw, dtype):
  return ops.nchw_to_nhwc(x_nchw.to(DEV, torch.float32).contiguous(), dtype)


def to_nchw(act):
  return ops.nhwc_to_nchw(act).cpu()


def rel_err(a, b):
  a = torch.as_tensor(a, dtype=torch.float64)
  b = torch.as_tensor(b, dtype=torch.float64)
  return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


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


def assert_close(a, b, tol, what=''):
  e = rel_err(a, b)
  assert e <= tol, '%s: max|a-b|/max|b| = %.3e > %.1e%s' % (what, e, tol, _where_bad(a, b, tol))


def bf16_round(t):
  return t.to(torch.bfloat16).to(torch.float32)


def quantize_like(t, dtype):
  """What the device sees after storing `t` in the compute dtype."""
  return bf16_round(t) if dtype == BF16 else t
