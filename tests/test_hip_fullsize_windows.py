"""GPU: element-wise parity at BASELINE.json's sizes (batch 4 @ 1024x512, and 2048x1024), where a whole-tensor oracle run
takes minutes.

The kernels that run at the bench size are not the ones the small parity cases of tests/test_hip_ops.py reach by default:
the row-streaming family (conv_rows / dgrad2_rows / head_rows / thin_rows / thin_in_rows), the halo kernel and the all-taps
weight gradients pick band heights, strip counts and pixel-range partitions from the problem size.  Their risk sits at strip
seams (multiples of 64 / 128 pixels), band seams (multiples of 16 / 32 / 64 rows), the image border and the batch boundary.
Here every layer runs ONCE at full size on the device and

  * forward and data gradient are compared element by element against torch-CPU (fp32 arithmetic on the same bf16-rounded
    operands) on full-width row bands -- top border, a band across row 64, a band across the middle seam, bottom border --
    of the first and the last image of the batch (>= 8 windows per layer; full width = every strip seam);
  * the weight gradient, a reduction over ALL pixels and images, is compared as the complete tensor against the sum of
    torch-CPU per-image weight gradients.

The ResnetBlock shape (32x64 pixels) is compared as complete tensors.  Criterion: hip_util.assert_close (max-norm AND
element-wise, RTOL[bf16]).  References: torch.nn.functional conv2d / conv_transpose2d as the reference's nn.Conv2d /
nn.ConvTranspose2d layers call them (ctu/models/pix2pixHD_networks/networks.py:204-262, 430-449).
"""
import zlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from jpdse_hip import BF16, PAD_ZERO, PAD_REFLECT, ACT_NONE
from jpdse_hip.ops import Act
from jpdse_hip.layers import HipConv2d
from hip_util import DEV, RTOL, assert_close


def _nchw(t_nhwc, C):
  """NHWC device slice -> fp32 NCHW on the CPU (logical channels only)."""
  return t_nhwc[..., :C].float().cpu().permute(0, 3, 1, 2).contiguous()


def _bands(OH):
  """Output-row bands [o0, o1): both borders, across row 64 (band seam of every band height <= 64) and across the middle."""
  if OH <= 48:
    return [(0, OH)]
  out = [(0, 12), (OH - 12, OH)]
  if OH >= 96:
    out.append((58, 70))
  mid = (OH // 2) // 2 * 2
  if OH >= 160:
    out.append((mid - 6, mid + 6))
  return out


def _conv_band(xb, w, k, st, pad, mode, pt, pb):
  xp = F.pad(xb, (pad, pad, pt, pb), mode='reflect' if mode == PAD_REFLECT else 'constant')
  return F.conv2d(xp, w, stride=st)


def _make(name, N, H, W, C, K, k, st, pad, mode, transposed):
  g = torch.Generator(device=DEV).manual_seed(zlib.crc32(name.encode()) % 1000)
  layer = HipConv2d(C, K, k, st, pad, mode, act=ACT_NONE, apply_bias=False, transposed=transposed, dtype=BF16, device=DEV)
  with torch.no_grad():
    layer.weight.copy_(torch.randn(layer.weight.shape, generator=g, device=DEV) * (1.0 / (C * k * k) ** 0.5))
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
  wq = layer.weight.detach().to(torch.bfloat16).float().cpu()      # what the packed panels hold
  return layer, x, y, dy, dx, wq


def _check_layer(name, N, H, W, C, K, k, st, pad, mode, transposed=False, wgrad_images=None):
  layer, x, y, dy, dx, wq = _make(name, N, H, W, C, K, k, st, pad, mode, transposed)
  tol = RTOL[BF16]
  OH, OW, Ky = y.H, y.W, y.C
  assert (y.t[..., Ky:] == 0).all() and (dx.t[..., C:] == 0).all(), 'padding lanes must stay zero'
  images = sorted({0, N - 1})
  nwin = 0
  if not transposed:
    for n in images:
      for (o0, o1) in _bands(OH):
        a0, a1 = o0 * st - pad, (o1 - 1) * st - pad + k
        pt, pb = max(0, -a0), max(0, a1 - H)
        a0c, a1c = max(0, a0), min(H, a1)
        xb = _nchw(x.t[n:n + 1, a0c:a1c], C).requires_grad_(True)
        yb = _conv_band(xb, wq, k, st, pad, mode, pt, pb)
        assert yb.shape[2] == o1 - o0
        assert_close(_nchw(y.t[n:n + 1, o0:o1], Ky), yb.detach(), tol, name + ' fwd band', detail='rows %d..%d image %d' % (o0, o1, n))
        yb.backward(_nchw(dy.t[n:n + 1, o0:o1], Ky))
        # input rows all of whose outputs lie inside the band (everything up to a true border)
        lo = 0 if o0 == 0 else a0c + (k - 1)
        hi = H if o1 == OH else a1c - (k - 1)
        if hi - lo >= 2:
          assert_close(_nchw(dx.t[n:n + 1, lo:hi], C), xb.grad[:, :, lo - a0c:hi - a0c], tol,
                       name + ' dgrad band', detail='rows %d..%d image %d' % (lo, hi, n))
        nwin += 1
  else:
    for n in images:
      for (o0, o1) in _bands(H):                     # bands of INPUT rows [a0, a1) -> output rows [2 a0, 2 a1)
        a0, a1 = o0, o1
        xb = _nchw(x.t[n:n + 1, a0:a1], C).requires_grad_(True)
        yb = F.conv_transpose2d(xb, wq, stride=2, padding=1, output_padding=1)
        ylo = 0 if a0 == 0 else 2
        yhi = yb.shape[2] if a1 == H else yb.shape[2] - 2
        assert_close(_nchw(y.t[n:n + 1, 2 * a0 + ylo:2 * a0 + yhi], Ky), yb.detach()[:, :, ylo:yhi], tol,
                     name + ' fwd band', detail='rows %d..%d image %d' % (2 * a0 + ylo, 2 * a0 + yhi, n))
        yb.backward(_nchw(dy.t[n:n + 1, 2 * a0:2 * a1], Ky))
        lo = 0 if a0 == 0 else 1
        hi = (a1 - a0) if a1 == H else (a1 - a0) - 1
        assert_close(_nchw(dx.t[n:n + 1, a0 + lo:a0 + hi], C), xb.grad[:, :, lo:hi], tol,
                     name + ' dgrad band', detail='rows %d..%d image %d' % (a0 + lo, a0 + hi, n))
        nwin += 1
  assert nwin >= 2
  # weight gradient: the complete tensor, summed over the images on the CPU
  wref = torch.zeros_like(wq)
  wl = wq.clone().requires_grad_(True)
  for n in (range(N) if wgrad_images is None else wgrad_images):
    xi = _nchw(x.t[n:n + 1], C)
    if transposed:
      yi = F.conv_transpose2d(xi, wl, stride=2, padding=1, output_padding=1)
    else:
      yi = _conv_band(xi, wl, k, st, pad, mode, pad, pad)
    (gw,) = torch.autograd.grad(yi, wl, _nchw(dy.t[n:n + 1], Ky))
    wref += gw
    del xi, yi, gw
  if wgrad_images is None:
    assert_close(layer.weight.grad.cpu(), wref, tol, name + ' wgrad (complete tensor, all images)')


# name, N, H, W, C, K, k, stride, pad, mode, transposed  -- the bench step's layers at 1024x512, batch 4 (PatchGAN: 8)
LAYERS_1024 = [
    ('g_first_7x7',      4, 512, 1024, 39,  64,  7, 1, 3, PAD_REFLECT, False),   # thin_fwd, wgrad_thin
    ('g_down_64_128',    4, 512, 1024, 64,  128, 3, 2, 1, PAD_ZERO,    False),   # conv_rows<2>, dgrad2_rows, wgrad_taps
    ('g_down_128_256',   4, 256, 512,  128, 256, 3, 2, 1, PAD_ZERO,    False),   # gemm_fast fwd / merged-phase dgrad
    ('g_up_convT_128_64', 4, 256, 512, 128, 64,  3, 2, 1, PAD_ZERO,    True),    # dgrad2_rows as the forward
    ('g_head_7x7',       4, 512, 1024, 64,  3,   7, 1, 3, PAD_REFLECT, False),   # head_rows<7>, thin_in_rows<7> + ring fold
    ('vgg_conv1_1',      4, 512, 1024, 3,   64,  3, 1, 1, PAD_ZERO,    False),   # thin_in_rows<3>, head_rows<3>
    ('vgg_conv1_2',      4, 512, 1024, 64,  64,  3, 1, 1, PAD_ZERO,    False),   # conv_rows<1,2>
    ('vgg_conv2_1',      4, 256, 512,  64,  128, 3, 1, 1, PAD_ZERO,    False),   # conv_rows<1,4>
    ('vgg_conv2_2',      4, 256, 512,  128, 128, 3, 1, 1, PAD_ZERO,    False),   # gemm_halo (zero padding, 2 slabs)
    ('d_layer0',         8, 512, 1024, 39,  64,  4, 2, 2, PAD_ZERO,    False),   # thin_rows
    ('d_layer1',         8, 257, 513,  64,  128, 4, 2, 2, PAD_ZERO,    False),   # odd sizes: ragged tiles
    ('d_layer3',         8, 65,  129,  256, 512, 4, 1, 2, PAD_ZERO,    False),   # 4x4 stride 1
    ('d_layer4',         8, 66,  130,  512, 1,   4, 1, 2, PAD_ZERO,    False),   # one output channel: thin1_dgrad / thin1_wgrad (thin_out1.h)
    # round 4: the grids the list above left out
    ('g_down_256_512',   4, 128, 256,  256, 512, 3, 2, 1, PAD_ZERO,    False),   # gemm_fast fwd, gemm_taps two-set dgrad, wgrad_taps
    ('g_down_512_1024',  4, 64,  128,  512, 1024, 3, 2, 1, PAD_ZERO,   False),
    ('g_up_convT_1024_512', 4, 32, 64, 1024, 512, 3, 2, 1, PAD_ZERO,   True),    # ConvTranspose forms of the same kernels
    ('g_up_convT_512_256',  4, 64, 128, 512, 256, 3, 2, 1, PAD_ZERO,   True),
    ('g_up_convT_256_128',  4, 128, 256, 256, 128, 3, 2, 1, PAD_ZERO,  True),
    ('d_layer2',         8, 129, 257,  128, 256, 4, 2, 2, PAD_ZERO,    False),   # 4x4 stride 2 on the odd 129 x 257 grid
    ('d_layer1_scale2',  8, 129, 257,  64,  128, 4, 2, 2, PAD_ZERO,    False),   # second PatchGAN scale (AvgPool'd input)
    ('vgg_conv4_2',      8, 64,  128,  512, 512, 3, 1, 1, PAD_ZERO,    False),   # batched [fake ; real] VGG pass: halo kernel, 8 slabs
    ('vgg_conv3_2',      8, 128, 256,  256, 256, 3, 1, 1, PAD_ZERO,    False),
    # LocalEnhancer-only layers at full resolution (BASELINE config 3; networks.py:160-175)
    ('local_resblock_64', 4, 256, 512, 64,  64,  3, 1, 1, PAD_REFLECT, False),   # single-slab halo + folded frame, wgrad_taps
    ('local_down_32_64', 4, 512, 1024, 32,  64,  3, 2, 1, PAD_ZERO,    False),
    ('local_convT_64_32', 4, 256, 512, 64,  32,  3, 2, 1, PAD_ZERO,    True),
    ('local_head_32_3',  4, 512, 1024, 32,  3,   7, 1, 3, PAD_REFLECT, False),
]


@pytest.mark.parametrize('case', LAYERS_1024, ids=[c[0] for c in LAYERS_1024])
def test_1024x512_windows_vs_torch_cpu_bf16(case):
  _check_layer(*case)


def test_1024x512_resblock_complete_tensors_vs_torch_cpu_bf16():
  """The headline kernel set (gemm_halo forward, halo + ring strips data gradient, wgrad_nine) at its bench shape: the
  complete forward, data-gradient and weight-gradient tensors (4 x 32 x 64 x 1024) against torch-CPU."""
  name, N, H, W, C = 'resblock_1024', 4, 32, 64, 1024
  layer, x, y, dy, dx, wq = _make(name, N, H, W, C, C, 3, 1, 1, PAD_REFLECT, False)
  tol = RTOL[BF16]
  xr = _nchw(x.t, C).requires_grad_(True)
  wl = wq.clone().requires_grad_(True)
  yr = _conv_band(xr, wl, 3, 1, 1, PAD_REFLECT, 1, 1)
  assert_close(_nchw(y.t, C), yr.detach(), tol, name + ' fwd (complete)')
  yr.backward(_nchw(dy.t, C))
  assert_close(_nchw(dx.t, C), xr.grad, tol, name + ' dgrad (complete)')
  assert_close(layer.weight.grad.cpu(), wl.grad, tol, name + ' wgrad (complete)')


def test_1024x512_local_trunk_complete_tensors_vs_torch_cpu_bf16():
  """BASELINE config 3's trunk: the 1024-channel ResnetBlock conv of the LocalEnhancer's coarse generator at 16 x 32 pixels,
  batch 4 -- nine-tap program forward, folded-frame data gradient, and (round 4) the row-pair form of the nine-tap weight
  gradient (wgrad_nine.h W32) -- as complete tensors against torch-CPU."""
  name, N, H, W, C = 'local_trunk_1024', 4, 16, 32, 1024
  layer, x, y, dy, dx, wq = _make(name, N, H, W, C, C, 3, 1, 1, PAD_REFLECT, False)
  tol = RTOL[BF16]
  xr = _nchw(x.t, C).requires_grad_(True)
  wl = wq.clone().requires_grad_(True)
  yr = _conv_band(xr, wl, 3, 1, 1, PAD_REFLECT, 1, 1)
  assert_close(_nchw(y.t, C), yr.detach(), tol, name + ' fwd (complete)')
  yr.backward(_nchw(dy.t, C))
  assert_close(_nchw(dx.t, C), xr.grad, tol, name + ' dgrad (complete)')
  assert_close(layer.weight.grad.cpu(), wl.grad, tol, name + ' wgrad (complete)')
  first = layer.weight.grad.clone()
  layer.bwd(layer.fwd(x)[1], dy, need_dx=False, need_dw=True)
  torch.cuda.synchronize()
  assert torch.equal(first, layer.weight.grad), 'the weight gradient must be bit-reproducible'


# 2048x1024 (BASELINE config 5), batch 1: forward / data-gradient windows; the weight gradient from image 0 only would be the
# same tensor as the device's (batch 1), so it is compared completely as well for the cheaper layers
LAYERS_2048 = [
    ('g_first_7x7@2k',   1, 1024, 2048, 39, 64,  7, 1, 3, PAD_REFLECT, False),
    ('g_down_64_128@2k', 1, 1024, 2048, 64, 128, 3, 2, 1, PAD_ZERO,    False),
    ('vgg_conv1_2@2k',   1, 1024, 2048, 64, 64,  3, 1, 1, PAD_ZERO,    False),
    ('g_head_7x7@2k',    1, 1024, 2048, 64, 3,   7, 1, 3, PAD_REFLECT, False),
]


@pytest.mark.parametrize('case', LAYERS_2048, ids=[c[0] for c in LAYERS_2048])
def test_2048x1024_windows_vs_torch_cpu_bf16(case):
  _check_layer(*case)


# ---- round 4: the FUSED variants the bench step actually runs, element by element at the bench shape -------------------------
# tests/test_hip_ops.py compares jpdse_conv_dgrad_fused / _fused_lrelu / jpdse_conv_fwd_pool / jpdse_conv_fwd_moments with
# torch-CPU at toy sizes only; at 1024x512 their kernels (conv_rows<1,2,true>, the halo kernel's mask / addend epilogue, the
# merged-phase fast kernel with the LeakyReLU gather, the pooled halo epilogue, the MOM epilogue) pick other strips and bands.
def _act_input(name, N, H, W, C, slope):
  """x = act(z) of a random pre-activation z (ReLU for slope 0, LeakyReLU(slope) otherwise), bf16."""
  g = torch.Generator(device=DEV).manual_seed(zlib.crc32(name.encode()) % 1000 + 31)
  x = Act.empty(N, H, W, C, BF16, DEV)
  x.t.zero_()
  z = torch.randn((N, H, W, C), generator=g, device=DEV)
  x.t[..., :C] = (torch.relu(z) if slope == 0.0 else F.leaky_relu(z, slope)).to(torch.bfloat16)
  return x, g


FUSED_DGRAD_1024 = [
    # name, N, H, W, C, K, k, stride, pad, mode, slope (0 = ReLU mask), addend
    ('vgg_conv1_2 dgrad_fused',   4, 512, 1024, 64,  64,  3, 1, 1, PAD_ZERO, 0.0, True),    # conv_rows<1,2,true>
    ('vgg_conv2_2 dgrad_fused',   4, 256, 512,  128, 128, 3, 1, 1, PAD_ZERO, 0.0, True),    # halo kernel epilogue: mask + fan-in addend
    ('vgg_conv3_2 dgrad_fused',   4, 128, 256,  256, 256, 3, 1, 1, PAD_ZERO, 0.0, False),   # mask only
    ('d_layer1 dgrad_fused_lrelu', 8, 257, 513, 64,  128, 4, 2, 2, PAD_ZERO, 0.2, False),   # loss_D backward: mask only, batch 8
    ('d_layer1 dgrad_fused_lrelu+addend', 4, 257, 513, 64, 128, 4, 2, 2, PAD_ZERO, 0.2, True),   # loss_G backward: + feature-matching tap
    ('d_layer2 dgrad_fused_lrelu+addend', 4, 129, 257, 128, 256, 4, 2, 2, PAD_ZERO, 0.2, True),
]


@pytest.mark.parametrize('case', FUSED_DGRAD_1024, ids=[c[0].replace(' ', '_') for c in FUSED_DGRAD_1024])
def test_1024x512_fused_dgrad_windows_vs_torch_cpu_bf16(case):
  """dx = (dgrad(dy) + addend) * act'(x) from ONE call (jpdse_conv_dgrad_fused / jpdse_conv_dgrad_fused_lrelu), compared element
  by element on full-width row bands with torch-CPU autograd through conv(x) plus the addend and the mask applied in fp32."""
  name, N, H, W, C, K, k, st, pad, mode, slope, with_addend = case
  x, g = _act_input(name, N, H, W, C, slope)
  layer = HipConv2d(C, K, k, st, pad, mode, act=ACT_NONE, apply_bias=False, dtype=BF16, device=DEV)
  with torch.no_grad():
    layer.weight.copy_(torch.randn(layer.weight.shape, generator=g, device=DEV) * (1.0 / (C * k * k) ** 0.5))
  y, ctx = layer.fwd(x)
  OH, Ky = y.H, y.C
  dy = y.empty_like()
  dy.t.zero_()
  dy.t[..., :Ky] = torch.randn(tuple(y.t.shape[:3]) + (Ky,), generator=g, device=DEV).to(torch.bfloat16)
  addend = None
  if with_addend:
    addend = x.empty_like()
    addend.t.zero_()
    addend.t[..., :C] = torch.randn((N, H, W, C), generator=g, device=DEV).to(torch.bfloat16)
  dx = layer.bwd(ctx, dy, need_dx=True, need_dw=False, relu_input=True, input_slope=slope, addend=addend)
  torch.cuda.synchronize()
  assert (dx.t[..., C:] == 0).all(), 'padding lanes must stay zero'
  wq = layer.weight.detach().to(torch.bfloat16).float().cpu()
  tol = RTOL[BF16]
  nwin = 0
  for n in sorted({0, N - 1}):
    for (o0, o1) in _bands(OH):
      a0, a1 = o0 * st - pad, (o1 - 1) * st - pad + k
      pt, pb = max(0, -a0), max(0, a1 - H)
      a0c, a1c = max(0, a0), min(H, a1)
      xb = _nchw(x.t[n:n + 1, a0c:a1c], C).requires_grad_(True)
      yb = _conv_band(xb, wq, k, st, pad, mode, pt, pb)
      yb.backward(_nchw(dy.t[n:n + 1, o0:o1], Ky))
      lo = 0 if o0 == 0 else a0c + (k - 1)
      hi = H if o1 == OH else a1c - (k - 1)
      if hi - lo < 2:
        continue
      want = xb.grad[:, :, lo - a0c:hi - a0c]
      if addend is not None:
        want = want + _nchw(addend.t[n:n + 1, lo:hi], C)
      xs = xb.detach()[:, :, lo - a0c:hi - a0c]
      want = want * torch.where(xs > 0, torch.ones_like(xs), torch.full_like(xs, slope))
      got = _nchw(dx.t[n:n + 1, lo:hi], C)
      assert_close(got, want, tol, name + ' band', detail='rows %d..%d image %d' % (lo, hi, n))
      if slope == 0.0:
        assert (got[xs <= 0] == 0).all(), name + ': ReLU-masked elements must be exactly zero'
      nwin += 1
  assert nwin >= 4


FWD_POOL_1024 = [
    ('vgg_conv2_2 fwd_pool', 8, 256, 512, 128, 128),     # batched [fake ; real] pass: batch 8
    ('vgg_conv3_4 fwd_pool', 8, 128, 256, 256, 256),
    ('vgg_conv4_4 fwd_pool', 8, 64,  128, 512, 512),
]


@pytest.mark.parametrize('case', FWD_POOL_1024, ids=[c[0].replace(' ', '_') for c in FWD_POOL_1024])
def test_1024x512_conv_fwd_pool_windows_vs_torch_cpu_bf16(case):
  """jpdse_conv_fwd_pool at the batched VGG19 shapes: conv + bias + ReLU on row bands against torch-CPU, and the pooled output
  against MaxPool2d(2, 2) of the device's own conv output, exactly (max is exact), over the COMPLETE tensor."""
  from jpdse_hip import ACT_RELU
  name, N, H, W, C, K = case
  g = torch.Generator(device=DEV).manual_seed(zlib.crc32(name.encode()) % 1000)
  layer = HipConv2d(C, K, 3, 1, 1, PAD_ZERO, act=ACT_RELU, dtype=BF16, device=DEV)
  with torch.no_grad():
    layer.weight.copy_(torch.randn(layer.weight.shape, generator=g, device=DEV) * (2.0 / (C * 9)) ** 0.5)
    layer.bias.copy_(torch.randn(K, generator=g, device=DEV) * 0.1)
  x = Act.empty(N, H, W, C, BF16, DEV)
  x.t.copy_(torch.relu(torch.randn((N, H, W, C), generator=g, device=DEV)).to(torch.bfloat16))
  y, yp, ctx = layer.fwd_pool(x)
  torch.cuda.synchronize()
  assert tuple(yp.t.shape) == (N, H // 2, W // 2, K)
  ref = y.t.view(N, H // 2, 2, W // 2, 2, K).amax(dim=(2, 4))
  assert torch.equal(yp.t, ref), name + ': pooled output is not the 2x2 maximum of the conv output'
  wq = layer.weight.detach().to(torch.bfloat16).float().cpu()
  b = layer.bias.detach().float().cpu()
  for n in (0, N - 1):
    for (o0, o1) in _bands(H):
      a0, a1 = o0 - 1, o1 + 1
      pt, pb = max(0, -a0), max(0, a1 - H)
      xb = _nchw(x.t[n:n + 1, max(0, a0):min(H, a1)], C)
      yb = torch.relu(_conv_band(xb, wq, 3, 1, 1, PAD_ZERO, pt, pb) + b.view(1, -1, 1, 1))
      assert_close(_nchw(y.t[n:n + 1, o0:o1], K), yb, RTOL[BF16], name + ' conv band', detail='rows %d..%d image %d' % (o0, o1, n))


def test_1024x512_resblock_fwd_moments_complete_vs_torch_cpu_bf16():
  """The MOM instantiation of the halo kernel at the ResnetBlock bench shape (jpdse_conv_fwd_moments): the conv output as a
  complete tensor against torch-CPU, bit-identical to the plain forward launch, and the (mean, rstd) its epilogue slots merge to
  against an fp64 pass over the stored tensor."""
  from jpdse_hip.layers import InstNormAct
  name, N, H, W, C = 'resblock_1024_mom', 4, 32, 64, 1024
  layer, x, y, dy, dx, wq = _make(name, N, H, W, C, C, 3, 1, 1, PAD_REFLECT, False)
  fused = layer.fwd_moments(x)
  assert fused is not None, 'the ResnetBlock conv is expected to take the fused-moment epilogue'
  y2, _c, mom, slots = fused
  yn, nctx = InstNormAct(ACT_NONE).fwd_from_moments(y2, mom, slots)
  torch.cuda.synchronize()
  assert torch.equal(y2.t, y.t), 'the MOM launch must produce the plain launch\'s output bit for bit'
  yr = _conv_band(_nchw(x.t, C), wq, 3, 1, 1, PAD_REFLECT, 1, 1)
  assert_close(_nchw(y2.t, C), yr, RTOL[BF16], name + ' fwd (complete)')
  stats = nctx.items[1]
  hf = y2.t[..., :C].double()
  mean_ref, var_ref = hf.mean(dim=(1, 2)), hf.var(dim=(1, 2), unbiased=False)
  assert_close(stats[:, :C, 0].cpu(), mean_ref.cpu(), 2e-5, name + ' fused mean vs fp64 of the stored tensor', elementwise=False)
  assert_close(stats[:, :C, 1].cpu(), (var_ref + 1e-5).rsqrt().cpu(), 5e-5, name + ' fused rstd vs fp64 of the stored tensor')
  yref = (hf - mean_ref[:, None, None]) * (var_ref + 1e-5).rsqrt()[:, None, None]
  assert_close(yn.t[..., :C].float().cpu(), yref.float().cpu(), RTOL[BF16], name + ' norm output from the fused moments')
