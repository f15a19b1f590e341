"""CPU-only checks of the boundary: the C-ABI library loads without a GPU and exports every
symbol include/jpdse.h declares; the host-side convolution planner (padding, "row-run" GEMM
addressing, stride-2 sub-pixel phases, reflect fold, filter packing) is emulated in numpy from
`jpdse_conv_plan_query` and compared with torch's conv2d / conv_transpose2d and their
gradients.  No device kernel is launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import jpdse_hip
from jpdse_hip import ConvDesc, F32, BF16, PAD_ZERO, PAD_REFLECT

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_header_symbols():
  L = jpdse_hip.lib()
  assert L.jpdse_version() == 2
  header = open(os.path.join(ROOT, 'include', 'jpdse.h')).read()
  declared = set(re.findall(r'\b(jpdse_[a-zA-Z0-9_]+)\s*\(', header))
  declared -= {'jpdse_conv_desc', 'jpdse_inorm_desc', 'jpdse_adam_entry'}
  assert len(declared) >= 38
  for name in sorted(declared):
    assert hasattr(L, name), 'header declares %s but the library does not export it' % name
  assert declared == set(jpdse_hip.SIGNATURES.keys())
  # the developer switch is NOT part of the shipped ABI: only libjpdse_hip_dev.so (include/jpdse_dev.h) has it
  assert not hasattr(L, 'jpdse_debug_set_fast_path'), 'the shipped library must not export the developer switch'
  dev_header = open(os.path.join(ROOT, 'include', 'jpdse_dev.h')).read()
  assert set(re.findall(r'\b(jpdse_[a-zA-Z0-9_]+)\s*\(', dev_header)) == set(jpdse_hip.DEV_SIGNATURES.keys())
  dev = ctypes.CDLL(jpdse_hip.DEV_LIB_PATH)
  for name in sorted(declared | set(jpdse_hip.DEV_SIGNATURES.keys())):
    assert hasattr(dev, name), 'developer build lacks %s' % name


def test_descriptor_validation_errors_are_reported():
  L = jpdse_hip.lib()
  d = ConvDesc(F32, 1, 8, 8, 4, 4, 3, 3, 3, 1, PAD_ZERO, 0, 0.2)     # stride 3 unsupported
  oh, ow = ctypes.c_int32(), ctypes.c_int32()
  assert L.jpdse_conv_out_shape(ctypes.byref(d), ctypes.byref(oh), ctypes.byref(ow)) == -1
  assert 'stride' in jpdse_hip.last_error()
  d = ConvDesc(F32, 1, 2, 2, 4, 4, 7, 7, 1, 3, PAD_REFLECT, 0, 0.2)  # reflect pad >= dim
  assert L.jpdse_conv_out_shape(ctypes.byref(d), ctypes.byref(oh), ctypes.byref(ow)) == -1
  with pytest.raises(jpdse_hip.JpdseError):
    jpdse_hip.check(-1, 'probe')


# ---- numpy emulation of the device-side semantics, driven by the C++ planner ------------------
def cpad(c):
  return (c + 7) & ~7


def plan(d):
  out = (ctypes.c_int32 * 54)()
  jpdse_hip.check(jpdse_hip.lib().jpdse_conv_plan_query(ctypes.byref(d), out, 54), 'plan_query')
  v = list(out)
  keys = 'Cs Ks Hp Wp OH OW Lk_fwd nph PT PB PL PR DH DW'.split()
  p = dict(zip(keys, v[:14]))
  p['phases'] = [dict(zip('qh qw Uh Uw i0h cnth i0w cntw Lk off'.split(), v[14 + 10 * i:24 + 10 * i]))
                 for i in range(p['nph'])]
  return p


def to_nhwc(x, Cs):
  n, c, h, w = x.shape
  out = np.zeros((n, h, w, Cs), dtype=np.float64)
  out[..., :c] = np.transpose(x, (0, 2, 3, 1))
  return out


def pad_nhwc(x, pt, pb, pl, pr, mode):
  return np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), mode='reflect' if mode == PAD_REFLECT else 'constant')


def gemm_rowrun(A_flat, pack, N, OH, OW, R, Lk, in_s, in_base, out_flat, out_s, out_base, Kvalid):
  """Y[m][k] = sum_{r,j} A[rowbase(m) + r*in_sr + j] * pack[k][r][j]  (the device kernel's contract)."""
  in_sn, in_sh, in_sw, in_sr = in_s
  out_sn, out_sh, out_sw = out_s
  for n in range(N):
    for oh in range(OH):
      for ow in range(OW):
        rb = in_base + n * in_sn + oh * in_sh + ow * in_sw
        acc = np.zeros(pack.shape[0])
        for r in range(R):
          run = A_flat[rb + r * in_sr: rb + r * in_sr + Lk]
          acc += pack[:, r, :] @ run
        ob = out_base + n * out_sn + oh * out_sh + ow * out_sw
        out_flat[ob: ob + pack.shape[0]] = acc


def emulate_fwd(d, p, x, w):
  Cs, Ks, Hp, Wp, OH, OW, Lk = (p[k] for k in 'Cs Ks Hp Wp OH OW Lk_fwd'.split())
  xp = pad_nhwc(to_nhwc(x, Cs), d.pad, d.pad, d.pad, d.pad, d.pad_mode)
  A = np.concatenate([xp.ravel(), np.zeros(4096)])
  pack = np.zeros((Ks, d.R, Lk))
  for s in range(d.S):
    pack[:d.K, :, s * Cs: s * Cs + d.C] = np.transpose(w[:, :, :, s], (0, 2, 1))   # w: [K,C,R,S]
  y = np.zeros(d.N * OH * OW * Ks)
  gemm_rowrun(A, pack, d.N, OH, OW, d.R, Lk, (Hp * Wp * Cs, d.stride * Wp * Cs, d.stride * Cs, Wp * Cs), 0,
              y, (OH * OW * Ks, OW * Ks, Ks), 0, d.K)
  return np.transpose(y.reshape(d.N, OH, OW, Ks)[..., :d.K], (0, 3, 1, 2))


def emulate_dgrad(d, p, dy, w):
  Cs, Ks, Hp, Wp, OH, OW = (p[k] for k in 'Cs Ks Hp Wp OH OW'.split())
  st = d.stride
  dyp = pad_nhwc(to_nhwc(dy, Ks), p['PT'], p['PB'], p['PL'], p['PR'], PAD_ZERO)
  assert dyp.shape[1] == p['DH'] and dyp.shape[2] == p['DW']
  A = np.concatenate([dyp.ravel(), np.zeros(4096)])
  refl = d.pad_mode == PAD_REFLECT
  dom_h, dom_w = (Hp, Wp) if refl else (d.H, d.W)
  out = np.full(d.N * dom_h * dom_w * Cs, np.nan)        # every element must be written exactly once
  for f in p['phases']:
    if f['cnth'] <= 0 or f['cntw'] <= 0:
      continue
    Lk = f['Lk']
    pack = np.zeros((Cs, f['Uh'], Lk))
    for up in range(f['Uh']):
      for wp in range(f['Uw']):
        r = f['qh'] + st * (f['Uh'] - 1 - up)
        s = f['qw'] + st * (f['Uw'] - 1 - wp)
        pack[:d.C, up, wp * Ks: wp * Ks + d.K] = w[:, :, r, s].T      # [C,K]
    in_base = ((f['i0h'] + p['PT'] - (f['Uh'] - 1)) * p['DW'] + (f['i0w'] + p['PL'] - (f['Uw'] - 1))) * Ks
    off = 0 if refl else d.pad
    out_base = ((st * f['i0h'] + f['qh'] - off) * dom_w + (st * f['i0w'] + f['qw'] - off)) * Cs
    gemm_rowrun(A, pack, d.N, f['cnth'], f['cntw'], f['Uh'], Lk,
                (p['DH'] * p['DW'] * Ks, p['DW'] * Ks, Ks, p['DW'] * Ks), in_base,
                out, (dom_h * dom_w * Cs, st * dom_w * Cs, st * Cs), out_base, d.C)
  assert not np.isnan(out).any(), 'dgrad phases left holes in the output'
  out = out.reshape(d.N, dom_h, dom_w, Cs)
  if refl:
    pd, H, W = d.pad, d.H, d.W
    dx = np.zeros((d.N, H, W, Cs))
    for h in range(H):
      hs = [h + pd] + ([pd - h] if 1 <= h <= pd else []) + ([pd + 2 * (H - 1) - h] if H - 1 - pd <= h <= H - 2 else [])
      for wv in range(W):
        ws = ([wv + pd] + ([pd - wv] if 1 <= wv <= pd else [])
              + ([pd + 2 * (W - 1) - wv] if W - 1 - pd <= wv <= W - 2 else []))
        for a in hs:
          for b in ws:
            dx[:, h, wv] += out[:, a, b]
    out = dx
  return np.transpose(out[..., :d.C], (0, 3, 1, 2))


def emulate_wgrad(d, p, x, dy):
  Cs, Ks, Hp, Wp, OH, OW = (p[k] for k in 'Cs Ks Hp Wp OH OW'.split())
  xp = pad_nhwc(to_nhwc(x, Cs), d.pad, d.pad, d.pad, d.pad, d.pad_mode)
  A = np.concatenate([xp.ravel(), np.zeros(4096)])
  dyn = to_nhwc(dy, Ks)
  run = d.S * Cs
  dw = np.zeros((d.K, d.R, run))
  for n in range(d.N):
    for oh in range(OH):
      for ow in range(OW):
        rb = n * Hp * Wp * Cs + oh * d.stride * Wp * Cs + ow * d.stride * Cs
        for r in range(d.R):
          dw[:, r, :] += np.outer(dyn[n, oh, ow, :d.K], A[rb + r * Wp * Cs: rb + r * Wp * Cs + run])
  dw = dw.reshape(d.K, d.R, d.S, Cs)[..., :d.C]
  return np.transpose(dw, (0, 3, 1, 2))     # [K,C,R,S]


CASES = [
    # N, H, W, C, K, R, stride, pad, mode          reference site
    (1, 6, 9, 5, 6, 7, 1, 3, PAD_REFLECT),       # ReflectionPad2d(3)+Conv7x7
    (2, 7, 8, 4, 9, 3, 2, 1, PAD_ZERO),          # Conv3x3 s2 p1 (odd height: unused last padded row)
    (1, 8, 6, 3, 4, 3, 2, 1, PAD_ZERO),          # ... even sizes: the ConvTranspose2d(op=1) geometry
    (2, 5, 6, 9, 10, 3, 1, 1, PAD_REFLECT),      # ResnetBlock conv
    (1, 9, 7, 6, 5, 4, 2, 2, PAD_ZERO),          # PatchGAN 4x4 s2 p2
    (1, 5, 6, 3, 2, 4, 1, 2, PAD_ZERO),          # PatchGAN 4x4 s1 p2
    (1, 6, 6, 3, 8, 3, 1, 1, PAD_ZERO),          # VGG conv
]


@pytest.mark.parametrize('case', CASES)
def test_conv_plan_matches_torch(case):
  N, H, W, C, K, R, st, pd, mode = case
  d = ConvDesc(F32, N, H, W, C, K, R, R, st, pd, mode, 0, 0.2)
  p = plan(d)
  g = torch.Generator().manual_seed(sum(case))
  x = torch.randn(N, C, H, W, generator=g, dtype=torch.float64, requires_grad=True)
  w = torch.randn(K, C, R, R, generator=g, dtype=torch.float64, requires_grad=True)
  xp = F.pad(x, (pd,) * 4, mode='reflect') if mode == PAD_REFLECT else F.pad(x, (pd,) * 4)
  y = F.conv2d(xp, w, stride=st)
  assert (p['OH'], p['OW']) == tuple(y.shape[2:])
  gy = torch.randn(y.shape, generator=g, dtype=torch.float64)
  y.backward(gy)
  np.testing.assert_allclose(emulate_fwd(d, p, x.detach().numpy(), w.detach().numpy()), y.detach().numpy(),
                             rtol=1e-10, atol=1e-10)
  np.testing.assert_allclose(emulate_dgrad(d, p, gy.numpy(), w.detach().numpy()), x.grad.numpy(),
                             rtol=1e-10, atol=1e-10)
  np.testing.assert_allclose(emulate_wgrad(d, p, x.detach().numpy(), gy.numpy()), w.grad.numpy(),
                             rtol=1e-10, atol=1e-10)


def test_conv_transpose_is_dgrad_of_underlying_conv():
  """nn.ConvTranspose2d(k3,s2,p1,op1) forward == dgrad of Conv2d(k3,s2,p1) whose KRSC master
  is the transposed conv's IOHW weight in channels_last memory (jpdse.h)."""
  Cin, Cout, Hin, Win = 6, 4, 3, 5
  g = torch.Generator().manual_seed(5)
  x = torch.randn(1, Cin, Hin, Win, generator=g, dtype=torch.float64)
  wt = torch.randn(Cin, Cout, 3, 3, generator=g, dtype=torch.float64)      # IOHW
  y = F.conv_transpose2d(x, wt, stride=2, padding=1, output_padding=1)
  d = ConvDesc(F32, 1, 2 * Hin, 2 * Win, Cout, Cin, 3, 3, 2, 1, PAD_ZERO, 0, 0.2)
  p = plan(d)
  assert (p['OH'], p['OW']) == (Hin, Win)
  # underlying conv weight [K=Cin][C=Cout][R][S] == wt itself
  np.testing.assert_allclose(emulate_dgrad(d, p, x.numpy(), wt.numpy()), y.numpy(), rtol=1e-10, atol=1e-10)


def test_workspace_and_pack_sizes_are_consistent():
  L = jpdse_hip.lib()
  for dt, es in ((F32, 4), (BF16, 2)):
    d = ConvDesc(dt, 2, 16, 32, 39, 64, 7, 7, 1, 3, PAD_REFLECT, 0, 0.2)
    p = plan(d)
    assert p['Cs'] == 40 and p['Ks'] == 64 and p['Lk_fwd'] % (64 // es) == 0 and p['Lk_fwd'] >= 7 * 40
    assert L.jpdse_conv_fwd_pack_size(ctypes.byref(d)) >= p['Ks'] * 7 * p['Lk_fwd'] * es
    need = 2 * p['Hp'] * p['Wp'] * p['Cs'] * es
    assert L.jpdse_conv_workspace_size(ctypes.byref(d)) >= need


def test_product_synthetic_inputs_match_the_oracle_generator():
  """bench.py's product leg must not import oracle/: ctu.utils.synthetic carries its own copy of the option
  defaults and of the synthetic x_dict generator; this pins the two copies to each other."""
  from ctu.utils import synthetic as prod
  from oracle.ctu_cpu import model as omodel
  a, b = vars(prod.default_opt(netG='local', ngf=32)), vars(omodel.default_opt(netG='local', ngf=32))
  assert a == b
  xa, xb = prod.synthetic_batch(2, 64, 96, seed=7), omodel.synthetic_batch(2, 64, 96, seed=7)
  assert xa.keys() == xb.keys()
  for k in xa:
    if torch.is_tensor(xa[k]):
      assert xa[k].dtype == xb[k].dtype and torch.equal(xa[k], xb[k]), k
    else:
      assert xa[k] == xb[k]


def test_bench_product_leg_does_not_import_the_oracle():
  src = open(os.path.join(os.path.dirname(__file__), '..', 'bench.py')).read()
  body = src[src.index('def main'):]
  assert 'oracle' not in body.replace('cpu_baseline', ''), 'only bench.py::cpu_baseline may touch oracle/'


def test_abi_v2_host_queries_without_a_gpu():
  """Host-only entry points added with ABI version 2 (include/jpdse.h): the norm-backward-sums slot count of a data-gradient
  kernel, and the per-phase weight-repack table with its fp32 / bf16 output flag.  None of them launches anything."""
  L = jpdse_hip.lib()
  assert L.jpdse_version() == 2
  assert ctypes.sizeof(jpdse_hip.PackEntry) == 2 * 8 + 16 * 4 + 2 * 8
  # ResnetBlock conv of the headline configuration (networks.py:246-252): one slot per 4-row x 64-pixel block of an image
  res = ConvDesc(BF16, 4, 64, 128, 1024, 1024, 3, 3, 1, 1, PAD_REFLECT, 0, 0.2)
  assert L.jpdse_conv_dgrad_nsum_slots(ctypes.byref(res)) == (64 // 4) * (128 // 64)
  for other in (ConvDesc(F32, 4, 64, 128, 1024, 1024, 3, 3, 1, 1, PAD_REFLECT, 0, 0.2),      # fp32: generic kernels
                ConvDesc(BF16, 4, 64, 128, 1024, 1024, 3, 3, 1, 1, PAD_ZERO, 0, 0.2),        # zero pad: not a ResnetBlock conv
                ConvDesc(BF16, 4, 64, 96, 1024, 1024, 3, 3, 1, 1, PAD_REFLECT, 0, 0.2),      # width not a multiple of 64
                ConvDesc(BF16, 4, 64, 128, 1024, 1024, 3, 3, 2, 1, PAD_ZERO, 0, 0.2)):       # strided
    assert L.jpdse_conv_dgrad_nsum_slots(ctypes.byref(other)) == 0
  # repack table: one entry per stride phase, panel pointers inside the caller's pack, fp32 panels flagged
  for dt in (F32, BF16):
    for st, nph in ((1, 1), (2, 4)):
      d = ConvDesc(dt, 1, 32, 64, 64, 128, 4 if st == 2 else 3, 4 if st == 2 else 3, st, 1, PAD_ZERO, 0, 0.2)
      ents = (jpdse_hip.PackEntry * 8)()
      size = L.jpdse_conv_dgrad_pack_size(ctypes.byref(d))
      base = 0x10000000
      n = L.jpdse_conv_pack_entries(ctypes.byref(d), ctypes.c_void_p(0x1000), ctypes.c_void_p(base), ents, 8)
      assert n == nph, (dt, st, n)
      for e in ents[:n]:
        assert e.w == 0x1000 and base <= e.out < base + size and e.blocks > 0
        assert e.out_f32 == (1 if dt == F32 else 0) and (e.K, e.C, e.st) == (128, 64, st)
      assert len({e.out for e in ents[:n]}) == n
      assert L.jpdse_conv_pack_entries(ctypes.byref(d), ctypes.c_void_p(0x1000), ctypes.c_void_p(base), ents, nph - 1) == -1
