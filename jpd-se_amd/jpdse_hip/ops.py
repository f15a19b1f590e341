"""Thin tensor-level wrappers over the C ABI.

torch is used for device memory (caching allocator) and the current HIP stream only; all
arithmetic happens inside libjpdse_hip.so.  Activations travel as `Act`: an NHWC tensor
[N,H,W,CPAD(C)] (fp32 or bf16) plus its logical channel count.
"""
import ctypes

import torch

from . import (lib, check, F32, BF16, ConvDesc, InormDesc, ACT_NONE, PAD_ZERO, PAD_REFLECT)


def cpad(c):
  return (c + 7) & ~7


def code_of(tdtype):
  if tdtype == torch.float32:
    return F32
  if tdtype == torch.bfloat16:
    return BF16
  raise TypeError('unsupported compute dtype %r' % (tdtype,))


def tdtype_of(code):
  return torch.bfloat16 if code == BF16 else torch.float32


def _p(t):
  return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def _stream():
  return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class Act(object):
  """NHWC activation: `t` has shape [N,H,W,CPAD(C)], padding lanes are zero."""
  __slots__ = ('t', 'C', 'nsums')

  def __init__(self, t, C):
    assert t.dim() == 4 and t.shape[3] == cpad(C) and t.is_contiguous(), (tuple(t.shape), C)
    self.t, self.C = t, C
    # a gradient tensor may carry the per-block sums of the InstanceNorm backward that consumes it, written by the epilogue
    # of the data-gradient kernel that produced it: (fp32 tensor [N][Cs][slots][2], slots, id of the norm's input tensor)
    self.nsums = None

  N = property(lambda s: s.t.shape[0])
  H = property(lambda s: s.t.shape[1])
  W = property(lambda s: s.t.shape[2])
  Cs = property(lambda s: s.t.shape[3])
  dtype = property(lambda s: code_of(s.t.dtype))

  def batch_slice(self, b0, b1):
    return Act(self.t[b0:b1], self.C)

  @staticmethod
  def empty(N, H, W, C, dtype_code, device):
    return Act(torch.empty((N, H, W, cpad(C)), dtype=tdtype_of(dtype_code), device=device), C)

  def empty_like(self):
    return Act(torch.empty_like(self.t), self.C)


# ---- shared workspace --------------------------------------------------------------------------
# One scratch arena per (device, stream): ops enqueued on one stream execute in order, so they may share scratch;
# a caller that runs a second torch stream (prefetch, eval overlap) gets its own arena instead of a silent race.
# Growing an arena hands the old buffer back to torch's caching allocator, which keeps it stream-ordered.
_ws = {}


def workspace(nbytes, device):
  index = device.index if device.index is not None else torch.cuda.current_device()
  key = (index, torch.cuda.current_stream(index).cuda_stream)
  cur = _ws.get(key)
  if cur is None or cur.numel() < nbytes:
    nbytes = int(nbytes * 1.25) + (1 << 20)
    cur = torch.empty(nbytes, dtype=torch.uint8, device=device)
    _ws[key] = cur
  return cur


# ---- convolution ------------------------------------------------------------------------------
def conv_desc(dtype, N, H, W, C, K, R, S, stride, pad, pad_mode, act=ACT_NONE, slope=0.2):
  return ConvDesc(dtype, N, H, W, C, K, R, S, stride, pad, pad_mode, act, slope)


def conv_out_shape(d):
  oh, ow = ctypes.c_int32(), ctypes.c_int32()
  check(lib().jpdse_conv_out_shape(ctypes.byref(d), ctypes.byref(oh), ctypes.byref(ow)), 'conv_out_shape')
  return oh.value, ow.value


def conv_pack(d, w_master, device):
  """fp32 KRSC master -> (fwd_pack, dgrad_pack) uint8 buffers in the compute dtype."""
  L = lib()
  nf, nd = L.jpdse_conv_fwd_pack_size(ctypes.byref(d)), L.jpdse_conv_dgrad_pack_size(ctypes.byref(d))
  fwd = torch.empty(nf, dtype=torch.uint8, device=device)
  dgr = torch.empty(nd, dtype=torch.uint8, device=device)
  check(L.jpdse_conv_pack_weights(ctypes.byref(d), _p(w_master), _p(fwd), _p(dgr), _stream()), 'conv_pack_weights')
  return fwd, dgr


def conv_pack_into(d, w_master, fwd, dgr):
  check(lib().jpdse_conv_pack_weights(ctypes.byref(d), _p(w_master), _p(fwd), _p(dgr), _stream()),
        'conv_pack_weights')


def _conv_ws(d, device):
  n = lib().jpdse_conv_workspace_size(ctypes.byref(d))
  return workspace(n, device), n


def conv_fwd(d, x, fwd_pack, bias, out=None):
  oh, ow = conv_out_shape(d)
  y = out if out is not None else Act.empty(d.N, oh, ow, d.K, d.dtype, x.t.device)
  ws, n = _conv_ws(d, x.t.device)
  check(lib().jpdse_conv_fwd(ctypes.byref(d), _p(x.t), _p(fwd_pack), _p(bias), _p(y.t), _p(ws), ws.numel(), _stream()),
        'conv_fwd')
  return y


def conv_fwd_pool(d, x, fwd_pack, bias):
  """(y, MaxPool2d(2, 2)(y)) in one call (jpdse_conv_fwd_pool)."""
  oh, ow = conv_out_shape(d)
  y = Act.empty(d.N, oh, ow, d.K, d.dtype, x.t.device)
  yp = Act.empty(d.N, oh // 2, ow // 2, d.K, d.dtype, x.t.device)
  ws, n = _conv_ws(d, x.t.device)
  check(lib().jpdse_conv_fwd_pool(ctypes.byref(d), _p(x.t), _p(fwd_pack), _p(bias), _p(y.t), _p(yp.t), _p(ws), ws.numel(),
                                  _stream()), 'conv_fwd_pool')
  return y, yp


def conv_dgrad(d, dy, dgrad_pack, relu_input=None, addend=None, mask_slope=0.0):
  """relu_input: the conv's own input x when it is a ReLU output -- dx is then masked where x <= 0
  (the ReLU backward fused into the GEMM epilogue); with mask_slope != 0 x is a LeakyReLU(mask_slope) output and
  dx is scaled by the slope there instead.  addend: another gradient w.r.t. the same tensor (skip connection,
  loss tap), summed in the epilogue: dx = (dgrad + addend) * mask."""
  dx = Act.empty(d.N, d.H, d.W, d.C, d.dtype, dy.t.device)
  ws, n = _conv_ws(d, dy.t.device)
  if relu_input is not None and mask_slope != 0.0:
    check(lib().jpdse_conv_dgrad_fused_lrelu(ctypes.byref(d), _p(dy.t), _p(dgrad_pack), _p(relu_input.t), float(mask_slope),
                                             _p(addend.t if addend is not None else None), _p(dx.t), _p(ws),
                                             ws.numel(), _stream()), 'conv_dgrad_fused_lrelu')
    return dx
  if relu_input is not None or addend is not None:
    check(lib().jpdse_conv_dgrad_fused(ctypes.byref(d), _p(dy.t), _p(dgrad_pack),
                                       _p(relu_input.t if relu_input is not None else None),
                                       _p(addend.t if addend is not None else None), _p(dx.t), _p(ws),
                                       ws.numel(), _stream()), 'conv_dgrad_fused')
    return dx
  check(lib().jpdse_conv_dgrad(ctypes.byref(d), _p(dy.t), _p(dgrad_pack), _p(dx.t), _p(ws), ws.numel(), _stream()),
        'conv_dgrad')
  return dx


def conv_dgrad_nsum_slots(d):
  """Blocks per image whose norm-backward sums the layer's data-gradient kernel can write (0: no such epilogue)."""
  return int(lib().jpdse_conv_dgrad_nsum_slots(ctypes.byref(d)))


def conv_dgrad_nsums(d, dy, dgrad_pack, slots, norm_x, norm_stats, norm_act, norm_slope, relu_input=None, addend=None):
  """conv_dgrad (+ mask / addend) whose dx feeds the backward of the InstanceNorm with input `norm_x` and statistics
  `norm_stats`: dx.nsums carries the per-block (sum dz, sum dz * yhat) slots for inorm_bwd (jpdse_conv_dgrad_fused_nsums)."""
  dx = Act.empty(d.N, d.H, d.W, d.C, d.dtype, dy.t.device)
  assert tuple(norm_x.t.shape) == tuple(dx.t.shape) and norm_x.dtype == dx.dtype
  sums = torch.empty((d.N, dx.Cs, slots, 2), dtype=torch.float32, device=dy.t.device)
  ws, n = _conv_ws(d, dy.t.device)
  check(lib().jpdse_conv_dgrad_fused_nsums(ctypes.byref(d), _p(dy.t), _p(dgrad_pack),
                                           _p(relu_input.t if relu_input is not None else None),
                                           _p(addend.t if addend is not None else None), _p(dx.t), _p(norm_x.t), _p(norm_stats),
                                           int(norm_act), float(norm_slope), _p(sums), _p(ws), ws.numel(), _stream()),
        'conv_dgrad_fused_nsums')
  dx.nsums = (sums, slots, norm_x.t.data_ptr())
  return dx


def conv_wgrad(d, x, dy, dw):
  """dw: fp32 tensor in the KRSC master layout, overwritten."""
  ws, n = _conv_ws(d, x.t.device)
  check(lib().jpdse_conv_wgrad(ctypes.byref(d), _p(x.t), _p(dy.t), _p(dw), _p(ws), ws.numel(), _stream()), 'conv_wgrad')


# ---- conv forward with the InstanceNorm moments fused into its epilogue -------------------------
def conv_moment_slots(d, transposed=False):
  """Blocks per image whose moments the layer's forward kernel writes (0: no such epilogue for this layer)."""
  fn = lib().jpdse_convT_moment_slots if transposed else lib().jpdse_conv_moment_slots
  return int(fn(ctypes.byref(d)))


def conv_fwd_moments(d, x, pack, slots, transposed=False):
  """y = conv(x) (no bias, no activation) and moments [N][CPAD(K)][slots][2] of y; for transposed=True `d` describes the
  underlying conv (its input is the ConvTranspose output) and `pack` is the data-gradient panel."""
  if transposed:
    y = Act.empty(d.N, d.H, d.W, d.C, d.dtype, x.t.device)
  else:
    oh, ow = conv_out_shape(d)
    y = Act.empty(d.N, oh, ow, d.K, d.dtype, x.t.device)
  mom = torch.empty((d.N, y.Cs, slots, 2), dtype=torch.float32, device=x.t.device)
  ws, n = _conv_ws(d, x.t.device)
  fn = lib().jpdse_convT_fwd_moments if transposed else lib().jpdse_conv_fwd_moments
  check(fn(ctypes.byref(d), _p(x.t), _p(pack), _p(y.t), _p(mom), _p(ws), ws.numel(), _stream()), 'conv_fwd_moments')
  return y, mom


def inorm_fwd_from_moments(x, mom, slots, act, slope=0.2, eps=1e-5, residual=None):
  d = InormDesc(x.dtype, x.N, x.H, x.W, x.C, act, slope, eps, 1 if residual is not None else 0)
  y = x.empty_like()
  stats = torch.empty((x.N, x.Cs, 2), dtype=torch.float32, device=x.t.device)
  check(lib().jpdse_inorm_fwd_from_moments(ctypes.byref(d), _p(x.t), _p(mom), slots,
                                           _p(residual.t if residual is not None else None), _p(y.t), _p(stats), _stream()),
        'inorm_fwd_from_moments')
  return y, stats


# ---- instance norm ----------------------------------------------------------------------------
def inorm_fwd(x, act, slope=0.2, eps=1e-5, residual=None):
  d = InormDesc(x.dtype, x.N, x.H, x.W, x.C, act, slope, eps, 1 if residual is not None else 0)
  y = x.empty_like()
  stats = torch.empty((x.N, x.Cs, 2), dtype=torch.float32, device=x.t.device)
  n = lib().jpdse_inorm_workspace_size(ctypes.byref(d))
  ws = workspace(n, x.t.device)
  check(lib().jpdse_inorm_fwd(ctypes.byref(d), _p(x.t), _p(residual.t if residual is not None else None), _p(y.t),
                              _p(stats), _p(ws), ws.numel(), _stream()), 'inorm_fwd')
  return y, stats


def inorm_bwd(x, stats, dy, act, slope=0.2, eps=1e-5):
  d = InormDesc(x.dtype, x.N, x.H, x.W, x.C, act, slope, eps, 0)
  dx = x.empty_like()
  n = lib().jpdse_inorm_workspace_size(ctypes.byref(d))
  ws = workspace(n, x.t.device)
  if dy.nsums is not None and dy.nsums[2] == x.t.data_ptr():
    # the sums were written by the epilogue of the kernel that produced dy, for THIS norm (same input tensor)
    sums, slots, _ = dy.nsums
    check(lib().jpdse_inorm_bwd_from_sums(ctypes.byref(d), _p(x.t), _p(stats), _p(dy.t), _p(sums), slots, _p(dx.t), _p(ws),
                                          ws.numel(), _stream()), 'inorm_bwd_from_sums')
    return dx
  check(lib().jpdse_inorm_bwd(ctypes.byref(d), _p(x.t), _p(stats), _p(dy.t), _p(dx.t), _p(ws), ws.numel(), _stream()),
        'inorm_bwd')
  return dx


# ---- pooling ----------------------------------------------------------------------------------
def avgpool3s2_fwd(x):
  y = Act.empty(x.N, (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1, x.C, x.dtype, x.t.device)
  check(lib().jpdse_avgpool3s2_fwd(x.dtype, x.N, x.H, x.W, x.C, _p(x.t), _p(y.t), _stream()), 'avgpool3s2_fwd')
  return y


def avgpool3s2_bwd(dy, H, W):
  dx = Act.empty(dy.N, H, W, dy.C, dy.dtype, dy.t.device)
  check(lib().jpdse_avgpool3s2_bwd(dy.dtype, dy.N, H, W, dy.C, _p(dy.t), _p(dx.t), _stream()), 'avgpool3s2_bwd')
  return dx


def maxpool2_fwd(x):
  y = Act.empty(x.N, x.H // 2, x.W // 2, x.C, x.dtype, x.t.device)
  check(lib().jpdse_maxpool2_fwd(x.dtype, x.N, x.H, x.W, x.C, _p(x.t), _p(y.t), _stream()), 'maxpool2_fwd')
  return y


def maxpool2_bwd(x, dy):
  dx = x.empty_like()
  check(lib().jpdse_maxpool2_bwd(x.dtype, x.N, x.H, x.W, x.C, _p(x.t), _p(dy.t), _p(dx.t), _stream()), 'maxpool2_bwd')
  return dx


# ---- elementwise -------------------------------------------------------------------------------
def act_bwd(y, dy, act, slope=0.2):
  dz = y.empty_like()
  check(lib().jpdse_act_bwd(y.dtype, y.t.numel(), act, slope, _p(y.t), _p(dy.t), _p(dz.t), _stream()), 'act_bwd')
  return dz


def add(a, b):
  out = a.empty_like()
  check(lib().jpdse_add(a.dtype, a.t.numel(), _p(a.t), _p(b.t), _p(out.t), _stream()), 'add')
  return out


def add_(a, b):
  check(lib().jpdse_add(a.dtype, a.t.numel(), _p(a.t), _p(b.t), _p(a.t), _stream()), 'add')
  return a


def copy_(src, dst):
  """dst = src (contiguous device tensors of equal byte size, a multiple of 16; views of larger tensors are fine)."""
  nb = src.numel() * src.element_size()
  assert src.is_contiguous() and dst.is_contiguous() and nb == dst.numel() * dst.element_size()
  check(lib().jpdse_copy(nb, _p(src), _p(dst), _stream()), 'copy')
  return dst


def zero_(t):
  check(lib().jpdse_zero(code_of(t.dtype), t.numel(), _p(t), _stream()), 'zero')
  return t


def cast_(src, dst):
  """dst = src converted fp32 <-> bf16 (flat device tensors of equal numel, a multiple of 8)."""
  assert src.numel() == dst.numel() and src.is_contiguous() and dst.is_contiguous()
  check(lib().jpdse_cast(code_of(src.dtype), code_of(dst.dtype), src.numel(), _p(src), _p(dst), _stream()), 'cast')
  return dst


def quant_loss(a, b, mean, std, mse, out):
  """Distortion between the uint8-quantised de-normalised images a, b (Acts, fp32 or bf16 each) into the fp32
  device slot `out` (0..255 scale): tensor2im + L1Loss/MSELoss of get_eval_loss, on the device."""
  assert a.t.shape == b.t.shape and a.C == b.C and len(mean) == a.C and len(std) == a.C
  ws = workspace(lib().jpdse_quant_loss_workspace_size(), a.t.device)
  arr = ctypes.c_double * a.C
  check(lib().jpdse_quant_loss(a.dtype, b.dtype, a.N * a.H * a.W, a.C, _p(a.t), _p(b.t), arr(*[float(v) for v in mean]),
                               arr(*[float(v) for v in std]), int(bool(mse)), _p(out), _p(ws), ws.numel(), _stream()),
        'quant_loss')


def channel_sum(dy, out):
  """out: fp32 [>= CPAD(C)] device tensor; receives the per-channel sums (bias gradient)."""
  npix = dy.N * dy.H * dy.W
  n = lib().jpdse_channel_sum_workspace_size(npix, dy.C)
  ws = workspace(n, dy.t.device)
  check(lib().jpdse_channel_sum(dy.dtype, npix, dy.C, _p(dy.t), _p(out), _p(ws), ws.numel(), _stream()), 'channel_sum')


def channel_copy(src, src_c0, dst, dst_c0, nch):
  assert src.t.shape[:3] == dst.t.shape[:3]
  npix = src.N * src.H * src.W
  check(lib().jpdse_channel_copy(src.dtype, npix, _p(src.t), src.Cs, src_c0, _p(dst.t), dst.Cs, dst_c0, nch, _stream()),
        'channel_copy')


def concat_channels(base, img, c0, out):
  """out = base with channels [c0, c0 + img.C) replaced by img (both NHWC Acts of the same N,H,W)."""
  npix = base.N * base.H * base.W
  check(lib().jpdse_concat_channels(base.dtype, npix, _p(base.t), base.t.shape[-1], _p(img.t), img.t.shape[-1], c0,
                                    img.C, _p(out.t), _stream()), 'concat_channels')
  return out


def nchw_to_nhwc(src, dtype_code, out=None):
  """fp32 NCHW device tensor -> Act (into `out` when given: an Act of the same shape, e.g. a batch slice of a larger one)."""
  assert src.dtype == torch.float32 and src.is_contiguous() and src.dim() == 4
  N, C, H, W = src.shape
  if out is None:
    out = Act.empty(N, H, W, C, dtype_code, src.device)
  assert tuple(out.t.shape) == (N, H, W, cpad(C)) and out.dtype == dtype_code
  check(lib().jpdse_nchw_to_nhwc(dtype_code, N, C, H, W, _p(src), _p(out.t), _stream()), 'nchw_to_nhwc')
  return out


def nhwc_to_nchw(x):
  out = torch.empty((x.N, x.C, x.H, x.W), dtype=torch.float32, device=x.t.device)
  check(lib().jpdse_nhwc_to_nchw(x.dtype, x.N, x.C, x.H, x.W, _p(x.t), _p(out), _stream()), 'nhwc_to_nchw')
  return out


def onehot_edge(label, instance, num_labels, total_c, dtype_code):
  """label fp32 [N,1,H,W], instance int64 [N,1,H,W] -> Act with `total_c` logical channels whose
  channels [0,num_labels) are the one-hot map and channel num_labels the edge map."""
  assert label.dtype == torch.float32 and instance.dtype == torch.int64
  assert label.is_contiguous() and instance.is_contiguous()
  N, _, H, W = label.shape
  out = Act.empty(N, H, W, total_c, dtype_code, label.device)
  check(lib().jpdse_onehot_edge(dtype_code, N, H, W, num_labels, _p(label), _p(instance), _p(out.t), out.Cs, _stream()),
        'onehot_edge')
  return out


def input_builder(label, instance, num_labels, dsts, imgs, c0):
  """One pass: every Act of `dsts` (same N,H,W,C) gets one-hot(label) | edge(instance) | its image of `imgs` (Act or None)
  at channels [c0, c0 + 3).  Replaces onehot_edge + one concat_channels per destination."""
  assert label.dtype == torch.float32 and instance.dtype == torch.int64 and label.is_contiguous() and instance.is_contiguous()
  assert 1 <= len(dsts) <= 3 and len(imgs) == len(dsts)
  N, _, H, W = label.shape
  d0 = dsts[0]
  present = [im for im in imgs if im is not None]
  img_cs, nch = (present[0].Cs, present[0].C) if present else (8, 3)
  for d, im in zip(dsts, imgs):
    assert tuple(d.t.shape) == tuple(d0.t.shape) and d.t.shape[:3] == (N, H, W) and d.dtype == d0.dtype
    assert im is None or (im.Cs == img_cs and im.C == nch and im.dtype == d0.dtype and im.t.shape[:3] == (N, H, W))
  arr = ctypes.c_void_p * len(dsts)
  check(lib().jpdse_input_builder(d0.dtype, N, H, W, num_labels, _p(label), _p(instance), len(dsts),
                                  arr(*[d.t.data_ptr() for d in dsts]), arr(*[(im.t.data_ptr() if im is not None else None) for im in imgs]),
                                  d0.Cs, img_cs, c0, nch, _stream()), 'input_builder')
  return dsts


def insert_channels(dst, img, c0):
  """dst[..., c0 : c0 + img.C] = img in place."""
  assert dst.t.shape[:3] == img.t.shape[:3] and dst.dtype == img.dtype
  npix = dst.N * dst.H * dst.W
  check(lib().jpdse_insert_channels(dst.dtype, npix, _p(dst.t), dst.Cs, _p(img.t), img.Cs, c0, img.C, _stream()), 'insert_channels')
  return dst


# ---- losses -------------------------------------------------------------------------------------
def _loss_ws(device):
  return workspace(lib().jpdse_loss_workspace_size(0), device)


# Deferred second stage of the loss reductions (jpdse_loss_finalize): inside `with deferred_loss_finals():` every loss
# forward leaves its block partials in its own region of a per-device buffer and ONE launch at the end of the block writes all
# the slots -- the train step computes 20 loss terms (pix2pixHD_pix2pixHD_model.py:205-221), each of which had its own 5-us final kernel.
_LOSS_TERMS_MAX = 64
import threading as _threading
_loss_tls = _threading.local()     # the open deferral list belongs to the thread that opened it (an eval pass on another
_loss_regions = {}                 # thread / stream must not share the training step's partial regions or its term list)


def _defer_list():
  return getattr(_loss_tls, 'defer', None)


class deferred_loss_finals(object):
  def __enter__(self):
    assert _defer_list() is None, 'deferred_loss_finals does not nest'
    _loss_tls.defer = []
    return self

  def __exit__(self, et, ev, tb):
    terms, _loss_tls.defer = _loss_tls.defer, None
    if et is None and terms:
      from . import LossTerm
      arr = (LossTerm * len(terms))()
      for i, (part, n, inv, out) in enumerate(terms):
        arr[i].partial, arr[i].n, arr[i].inv_count, arr[i].out = part, n, inv, out
      check(lib().jpdse_loss_finalize(arr, len(terms), _stream()), 'loss_finalize')
    return False


def _loss_target(out, device, work_items, count):
  """(ws tensor, out pointer or None): the shared loss workspace and the caller's slot, or -- deferred -- a private region and NULL."""
  defer = _defer_list()
  if defer is None:
    return _loss_ws(device), _p(out)
  per = lib().jpdse_loss_workspace_size(0) // 4
  index = device.index if device.index is not None else torch.cuda.current_device()
  key = (index, torch.cuda.current_stream(index).cuda_stream)       # one region buffer per (device, stream), as workspace()
  buf = _loss_regions.get(key)
  if buf is None:
    buf = _loss_regions[key] = torch.empty(_LOSS_TERMS_MAX * per, dtype=torch.float32, device=device)
  i = len(defer)
  if i >= _LOSS_TERMS_MAX:
    return _loss_ws(device), _p(out)                  # more terms than regions: this one finishes on its own
  region = buf[i * per:(i + 1) * per]
  defer.append((region.data_ptr(), int(lib().jpdse_loss_partial_count(int(work_items))), 1.0 / float(count), out.data_ptr()))
  return region, None


def _vec_items(a):
  return a.t.numel() // (4 if a.dtype == F32 else 8)


def l1_fwd(a, b, out):
  """out: fp32 device scalar slot (1-element view); mean over LOGICAL elements."""
  count = a.N * a.H * a.W * a.C
  ws, o = _loss_target(out, a.t.device, _vec_items(a), count)
  check(lib().jpdse_l1_fwd(a.dtype, a.t.numel(), count, _p(a.t), _p(b.t), o, _p(ws), ws.numel() * ws.element_size(), _stream()), 'l1_fwd')


def l1_fwd_bwd(a, b, out, scale, relu_a=False):
  """Loss value into `out` and its gradient w.r.t. a (times `scale`) in one pass; returns the gradient."""
  da = a.empty_like()
  count = a.N * a.H * a.W * a.C
  ws, o = _loss_target(out, a.t.device, _vec_items(a), count)
  check(lib().jpdse_l1_fwd_bwd(a.dtype, a.t.numel(), count, _p(a.t), _p(b.t), o, float(scale), int(relu_a),
                               _p(da.t), _p(ws), ws.numel() * ws.element_size(), _stream()), 'l1_fwd_bwd')
  return da


def l1_bwd(a, b, gout, scale, relu_a=False):
  """relu_a: `a` is a ReLU output; return the gradient w.r.t. its pre-activation."""
  da = a.empty_like()
  count = a.N * a.H * a.W * a.C
  fn = lib().jpdse_l1_bwd_relu if relu_a else lib().jpdse_l1_bwd
  check(fn(a.dtype, a.t.numel(), count, _p(a.t), _p(b.t), _p(gout), scale, _p(da.t), _stream()), 'l1_bwd')
  return da


def mse_fwd(a, b, out):
  count = a.N * a.H * a.W * a.C
  ws, o = _loss_target(out, a.t.device, _vec_items(a), count)
  check(lib().jpdse_mse_fwd(a.dtype, a.t.numel(), count, _p(a.t), _p(b.t), o, _p(ws), ws.numel() * ws.element_size(), _stream()), 'mse_fwd')


def mse_bwd(a, b, gout, scale):
  da = a.empty_like()
  count = a.N * a.H * a.W * a.C
  check(lib().jpdse_mse_bwd(a.dtype, a.t.numel(), count, _p(a.t), _p(b.t), _p(gout), scale, _p(da.t), _stream()), 'mse_bwd')
  return da


def mse_const_fwd(x, target, out):
  assert x.C == 1
  npix = x.N * x.H * x.W
  ws, o = _loss_target(out, x.t.device, npix, npix)
  check(lib().jpdse_mse_const_fwd(x.dtype, npix, x.Cs, target, _p(x.t), o, _p(ws), ws.numel() * ws.element_size(), _stream()),
        'mse_const_fwd')


def mse_const_bwd(x, target, gout, scale, out=None):
  dx = out if out is not None else x.empty_like()
  npix = x.N * x.H * x.W
  check(lib().jpdse_mse_const_bwd(x.dtype, npix, x.Cs, target, _p(x.t), _p(gout), scale, _p(dx.t), _stream()),
        'mse_const_bwd')
  return dx
