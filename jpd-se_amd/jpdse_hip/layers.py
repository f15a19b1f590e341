"""Layer objects with explicit forward/backward over the C ABI (no autograd on the hot path).

Each layer exposes
    fwd(x: Act) -> (y: Act, ctx)          ctx = what backward needs (batch-sliceable)
    bwd(ctx, dy: Act, need_dx, need_dw) -> dx | None
Parameters are torch Parameters whose logical shapes/keys equal the reference checkpoint
layout while their memory is channels_last (= KRSC, the ABI's master
layout), so `state_dict()` round-trips with reference `net_G.pth` / `net_D.pth` files.
Weight gradients are written by the kernels straight into persistent `.grad` buffers.
"""
import ctypes
import math

import torch
import torch.nn as nn

from . import ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, PAD_ZERO, PAD_REFLECT, F32, BF16
from . import ops
from .ops import Act

_weights_epoch = [0]


def lib_pack_size(d):
  from . import lib
  return int(lib().jpdse_conv_dgrad_pack_size(ctypes.byref(d)))


def bump_weights_epoch():
  """Call after any in-place update of master weights done outside torch's version counter
  (the fused Adam kernel): packed GEMM panels are rebuilt lazily on next use."""
  _weights_epoch[0] += 1


def current_weights_epoch():
  return _weights_epoch[0]


class Ctx(object):
  """Saved tensors of one layer call; every entry is batch-first so a sub-batch can be
  back-propagated on its own (used for the fake half of the batched discriminator pass)."""
  __slots__ = ('items',)

  def __init__(self, *items):
    self.items = items

  def slice(self, b0, b1):
    out = []
    for it in self.items:
      if isinstance(it, Act):
        out.append(it.batch_slice(b0, b1))
      elif torch.is_tensor(it):
        out.append(it[b0:b1])
      elif isinstance(it, Ctx):
        out.append(it.slice(b0, b1))
      elif isinstance(it, (list, tuple)):
        out.append(type(it)(c.slice(b0, b1) if isinstance(c, Ctx) else c for c in it))
      else:
        out.append(it)
    return Ctx(*out)


class HipConv2d(nn.Module):
  """nn.Conv2d / nn.ConvTranspose2d with the padding layer and the activation folded in."""

  def __init__(self, cin, cout, k, stride=1, pad=0, pad_mode=PAD_ZERO, act=ACT_NONE, slope=0.2,
               apply_bias=True, transposed=False, dtype=F32, device=None):
    super(HipConv2d, self).__init__()
    self.cin, self.cout, self.k = cin, cout, k
    self.stride, self.pad, self.pad_mode = stride, pad, pad_mode
    self.act, self.slope = act, slope
    self.apply_bias = apply_bias      # False: conv feeds an affine-less InstanceNorm (bias is dead)
    self.transposed = transposed
    self.cdtype = dtype
    if transposed:
      assert stride == 2 and pad == 1 and k == 3 and act == ACT_NONE and not apply_bias
      shape = (cin, cout, k, k)       # torch IOHW
    else:
      shape = (cout, cin, k, k)
    w = torch.empty(shape, device=device).contiguous(memory_format=torch.channels_last)
    w.normal_(0.0, 0.02)              # weights_init (networks.py:19-25)
    fan_in = shape[1] * k * k
    bound = 1.0 / math.sqrt(fan_in)
    b = torch.empty(cout, device=device).uniform_(-bound, bound)
    self.weight = nn.Parameter(w)
    self.bias = nn.Parameter(b)
    self._packs = None
    self._pack_key = None
    self._bias_grad_store = None
    self.grad_ready_hook = None       # callable(param) fired right after a gradient is written

  # -- descriptors ---------------------------------------------------------------------------
  def _desc(self, N, H, W):
    """H, W: spatial size of the underlying Conv2d's INPUT."""
    if self.transposed:
      return ops.conv_desc(self.cdtype, N, H, W, self.cout, self.cin, self.k, self.k, 2, 1, PAD_ZERO)
    return ops.conv_desc(self.cdtype, N, H, W, self.cin, self.cout, self.k, self.k, self.stride, self.pad,
                         self.pad_mode, self.act, self.slope)

  def _master(self):
    w = self.weight
    assert w.dtype == torch.float32 and w.permute(0, 2, 3, 1).is_contiguous(), \
        'master weights must be fp32 channels_last (KRSC)'
    return w

  def _plain_fwd_panel(self):
    """True when the forward panel is an element-for-element bf16 cast of the KRSC master (no channel or
    chunk padding, no Toeplitz extra) -- then the fused Adam kernel can write it."""
    C, K = (self.cout, self.cin) if self.transposed else (self.cin, self.cout)
    return self.cdtype == BF16 and C % 64 == 0 and K % 8 == 0 and K > 8     # thin inputs / heads carry extra panels

  def _fp32_alias_fwd_panel(self):
    """fp32 layers whose forward panel [Ks][R][S*Cs] would be an element-for-element copy of the KRSC master (no channel or
    chunk padding): the kernels read the master itself -- no copy after each optimizer step (round 4: 32 such copies per step
    cost BASELINE config 2 0.4 ms of 24)."""
    C, K = (self.cout, self.cin) if self.transposed else (self.cin, self.cout)
    return self.cdtype == F32 and C % 8 == 0 and K % 8 == 0 and (self.k * C) % 16 == 0

  def packs(self):
    w = self._master()
    wver = getattr(w, '_jpdse_wver', 0)        # bumped by FusedAdam for the tensors it updated
    key = (w._version, _weights_epoch[0], w.data_ptr(), self.cdtype, wver)
    if self._pack_key != key and self._fp32_alias_fwd_panel():
      d = self._desc(1, 64, 64)
      if self._packs is None or self._pack_key[3] != self.cdtype or self._packs[0].data_ptr() != w.data_ptr():
        nd = lib_pack_size(d)
        self._packs = (w.detach(), torch.empty(nd, dtype=torch.uint8, device=w.device))
      ops.conv_pack_into(d, w, None, self._packs[1])        # the data-gradient panels only
      self._pack_key = key
      return self._packs
    if self._pack_key != key:
      d = self._desc(1, 64, 64)   # pack layout does not depend on N,H,W
      if self._packs is None or self._pack_key[3] != self.cdtype:
        self._packs = ops.conv_pack(d, w, w.device)
        w._jpdse_cast_out = self._packs[0] if self._plain_fwd_panel() else None
      elif (getattr(w, '_jpdse_cast_out', None) is self._packs[0] and
            getattr(w, '_jpdse_cast_wver', None) == wver and self._pack_key[:4] == key[:4]):
        ops.conv_pack_into(d, w, None, self._packs[1])      # forward panel written by the Adam kernel
      else:
        ops.conv_pack_into(d, w, self._packs[0], self._packs[1])
      self._pack_key = key
    return self._packs

  def _pack_key_now(self):
    w = self._master()
    return (w._version, _weights_epoch[0], w.data_ptr(), self.cdtype, getattr(w, '_jpdse_wver', 0))

  def batched_pack_entries(self):
    """Table entries for PackBatcher, or None when this layer needs its own pack call right now (first use,
    fp32, forward panel not written by Adam, weights replaced, padded panel rows)."""
    w = self._master()
    if self._packs is None or self._pack_key is None:
      return None
    if self.cdtype != BF16 and not (self._fp32_alias_fwd_panel() and self._packs[0].data_ptr() == w.data_ptr()):
      return None
    key = self._pack_key_now()
    if key == self._pack_key:
      return []                                     # nothing to do
    if self.cdtype == BF16 and not (getattr(w, '_jpdse_cast_out', None) is self._packs[0] and
                                    getattr(w, '_jpdse_cast_wver', None) == key[4] and self._pack_key[:4] == key[:4]):
      return None
    if self.cdtype != BF16 and self._pack_key[:4] != key[:4]:
      return None                                   # weights replaced (load_state_dict): the lazy per-layer path
    if getattr(self, '_pack_entries_cache', None) is None or self._pack_entries_cache[0] != (w.data_ptr(), self._packs[1].data_ptr()):
      from . import lib, PackEntry
      buf = (PackEntry * 4)()
      d = self._desc(1, 64, 64)
      n = lib().jpdse_conv_pack_entries(ctypes.byref(d), ctypes.c_void_p(w.data_ptr()),
                                        ctypes.c_void_p(self._packs[1].data_ptr()), buf, 4)
      self._pack_entries_cache = ((w.data_ptr(), self._packs[1].data_ptr()), [buf[i] for i in range(n)] if n >= 0 else None)
    return self._pack_entries_cache[1]

  def _wgrad_buffer(self):
    w = self.weight
    if w.grad is None or w.grad.data_ptr() == 0 or not w.grad.permute(0, 2, 3, 1).is_contiguous():
      w.grad = torch.zeros_like(w, memory_format=torch.preserve_format)
    return w.grad

  def _bgrad_buffer(self):
    if self._bias_grad_store is None or self._bias_grad_store.device != self.bias.device:
      self._bias_grad_store = torch.zeros(ops.cpad(self.cout), dtype=torch.float32, device=self.bias.device)
      self.bias.grad = self._bias_grad_store[:self.cout]
    elif self.bias.grad is None:
      self.bias.grad = self._bias_grad_store[:self.cout]
    return self._bias_grad_store

  def ensure_grads(self):
    if self.weight.requires_grad:
      self._wgrad_buffer()
      self._bgrad_buffer()

  # -- forward / backward ---------------------------------------------------------------------
  def fwd(self, x):
    fwd_pack, dgrad_pack = self.packs()
    if self.transposed:
      d = self._desc(x.N, 2 * x.H, 2 * x.W)
      y = ops.conv_dgrad(d, x, dgrad_pack)
      return y, Ctx(x)
    d = self._desc(x.N, x.H, x.W)
    y = ops.conv_fwd(d, x, fwd_pack, self.bias if self.apply_bias else None)
    return y, Ctx(x, y if self.act != ACT_NONE else None)

  def fwd_pool(self, x):
    """(y, MaxPool2d(2, 2)(y), ctx): the conv -> ReLU -> pool chain of VGG19 in one call."""
    assert not self.transposed
    fwd_pack, _ = self.packs()
    d = self._desc(x.N, x.H, x.W)
    y, yp = ops.conv_fwd_pool(d, x, fwd_pack, self.bias if self.apply_bias else None)
    return y, yp, Ctx(x, y if self.act != ACT_NONE else None)

  def fwd_moments(self, x):
    """Forward for a conv whose output goes straight into an affine-less InstanceNorm: (y, ctx, moments, slots) with the
    norm's moment pass fused into the conv epilogue (jpdse_conv_fwd_moments), or None when this layer's kernel has no such
    epilogue (or the layer applies a bias / activation)."""
    if self.apply_bias or self.act != ACT_NONE or self.cdtype != BF16:
      return None
    d = self._desc(x.N, 2 * x.H, 2 * x.W) if self.transposed else self._desc(x.N, x.H, x.W)
    from . import binding_epoch
    key = (binding_epoch(), x.N, x.H, x.W)       # the answer depends on which library / kernel-selection mode is bound
    cache = self.__dict__.setdefault('_moment_slots', {})
    if key not in cache:
      cache[key] = ops.conv_moment_slots(d, self.transposed)
    slots = cache[key]
    if slots <= 0:
      return None
    fwd_pack, dgrad_pack = self.packs()
    y, mom = ops.conv_fwd_moments(d, x, dgrad_pack if self.transposed else fwd_pack, slots, self.transposed)
    return y, (Ctx(x) if self.transposed else Ctx(x, None)), mom, slots

  def nsum_slots(self, N, H, W):
    """Slots per image of the norm-backward sums this layer's data-gradient epilogue can write (0: none), cached per shape."""
    if self.transposed or self.cdtype != BF16:
      return 0
    from . import binding_epoch
    key = (binding_epoch(), N, H, W)
    cache = self.__dict__.setdefault('_nsum_slots', {})
    if key not in cache:
      cache[key] = ops.conv_dgrad_nsum_slots(self._desc(N, H, W))
    return cache[key]

  def bwd(self, ctx, dy, need_dx=True, need_dw=True, dy_is_dz=False, relu_input=False, addend=None, input_slope=0.0, sink=None):
    """dy_is_dz: dy is already the gradient w.r.t. the PRE-activation (the caller fused this layer's
    activation backward upstream); relu_input: the layer's input is a ReLU output (LeakyReLU(input_slope) when
    input_slope != 0) and the returned dx is wanted w.r.t. that activation's pre-activation (mask fused into the
    data-gradient epilogue); addend: another gradient w.r.t. the layer's input, summed into dx in the same epilogue;
    sink = (norm input Act, stats, act, slope): dx is the gradient w.r.t. the OUTPUT of that InstanceNorm (+ activation) -- when
    this layer's data-gradient kernel can, its epilogue also writes the sums of that norm's backward (dx.nsums)."""
    fwd_pack, dgrad_pack = self.packs()
    need_dw = need_dw and self.weight.requires_grad
    if self.transposed:
      (x,) = ctx.items
      d = self._desc(x.N, 2 * x.H, 2 * x.W)
      if need_dw:
        ops.conv_wgrad(d, dy, x, self._wgrad_buffer())
        self._bgrad_buffer()
        self._fire()
      return ops.conv_fwd(d, dy, fwd_pack, None) if need_dx else None
    x, y = ctx.items
    d = self._desc(x.N, x.H, x.W)
    dz = dy if (self.act == ACT_NONE or dy_is_dz) else ops.act_bwd(y, dy, self.act, self.slope)
    if need_dw:
      ops.conv_wgrad(d, x, dz, self._wgrad_buffer())
      store = self._bgrad_buffer()
      if self.apply_bias:
        ops.channel_sum(dz, store)          # writes CPAD(K) sums into the layer's private store
        if self.bias.grad.data_ptr() != store.data_ptr():
          self.bias.grad.copy_(store[:self.cout])   # .grad re-homed into a DDP bucket view
      self._fire()
    if not need_dx:
      return None
    if sink is not None and input_slope == 0.0:
      slots = self.nsum_slots(x.N, x.H, x.W)
      if slots > 0:
        sx, sstats, sact, sslope = sink
        return ops.conv_dgrad_nsums(d, dz, dgrad_pack, slots, sx, sstats, sact, sslope,
                                    relu_input=x if relu_input else None, addend=addend)
    return ops.conv_dgrad(d, dz, dgrad_pack, relu_input=x if relu_input else None, addend=addend, mask_slope=input_slope)

  def bwd_input_slice(self, ctx, dy, c0, c1, dy_is_dz=False):
    """Data gradient w.r.t. input channels [c0, c1) only (no weight gradient): the conv restricted to those input
    channels has the filter w[:, c0:c1], and its data gradient is that slice of the full one.  Used where only part
    of a concatenated input needs a gradient (PatchGAN layer 0: 3 image channels of the 39-channel input -- the
    full data gradient is 13x the work and 5x the bytes)."""
    assert not self.transposed
    x, y = ctx.items
    dz = dy if (self.act == ACT_NONE or dy_is_dz) else ops.act_bwd(y, dy, self.act, self.slope)
    w = self._master()
    key = (w._version, _weights_epoch[0], w.data_ptr(), self.cdtype, getattr(w, '_jpdse_wver', 0), c0, c1)
    d = ops.conv_desc(self.cdtype, x.N, x.H, x.W, c1 - c0, self.cout, self.k, self.k, self.stride, self.pad,
                      self.pad_mode, ACT_NONE, self.slope)
    if getattr(self, '_slice_key', None) != key:
      ws = w.detach()[:, c0:c1].contiguous(memory_format=torch.channels_last)
      # same dtype and channel range as last time: re-pack into the existing panels (the weight version alone changes every step)
      if getattr(self, '_slice_packs', None) is None or (self._slice_key[3], self._slice_key[5:]) != (key[3], key[5:]):
        self._slice_packs = ops.conv_pack(d, ws, w.device)
      else:
        ops.conv_pack_into(d, ws, self._slice_packs[0], self._slice_packs[1])
      self._slice_key = key
    return ops.conv_dgrad(d, dz, self._slice_packs[1])

  def _fire(self):
    if self.grad_ready_hook is not None:
      self.grad_ready_hook(self)


class PackBatcher(object):
  """Re-packs the data-gradient panels of all conv layers of one network in ONE launch right after its
  optimizer step (jpdse_conv_pack_run); layers it cannot cover keep their lazy per-layer pack."""

  def __init__(self, net):
    self.net = net
    self._sig = None
    self._table = None

  def run(self):
    from . import lib, check, PackEntry
    layers, entries = [], []
    for m in self.net.modules():
      if isinstance(m, HipConv2d):
        e = m.batched_pack_entries()
        if e:
          layers.append(m)
          entries.extend(e)
    if not entries:
      return 0
    sig = tuple((e.w, e.out) for e in entries)
    if sig != self._sig:
      arr = (PackEntry * len(entries))()
      block0 = 0
      for i, e in enumerate(entries):
        arr[i] = e
        arr[i].block0 = block0
        block0 += e.blocks
      host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
      self._table = (host.to(layers[0].weight.device), len(entries), block0)
      self._sig = sig
    dev, n, total = self._table
    check(lib().jpdse_conv_pack_run(ctypes.c_void_p(dev.data_ptr()), n, total,
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'conv_pack_run')
    for m in layers:
      m._pack_key = m._pack_key_now()
    return len(layers)


class InstNormAct(object):
  """InstanceNorm2d(affine=False, eps=1e-5) + {none, ReLU, LeakyReLU(0.2)} (+ residual)."""

  def __init__(self, act=ACT_NONE, slope=0.2, eps=1e-5):
    self.act, self.slope, self.eps = act, slope, eps

  def fwd(self, x, residual=None):
    y, stats = ops.inorm_fwd(x, self.act, self.slope, self.eps, residual)
    return y, Ctx(x, stats)

  def fwd_from_moments(self, x, mom, slots, residual=None):
    y, stats = ops.inorm_fwd_from_moments(x, mom, slots, self.act, self.slope, self.eps, residual)
    return y, Ctx(x, stats)

  def bwd(self, ctx, dy, need_dx=True, need_dw=True):
    x, stats = ctx.items
    return ops.inorm_bwd(x, stats, dy, self.act, self.slope, self.eps)     # takes dy.nsums when the producer of dy wrote them

  def sink(self, ctx):
    """What the producer of this norm's dy needs to write the backward sums in its epilogue (HipConv2d.bwd sink=)."""
    x, stats = ctx.items
    return (x, stats, self.act, self.slope)


class ConvNormAct(object):
  """[pad] conv -> InstanceNorm -> activation: one stage of the generator trunk / PatchGAN.
  A plain object (not an nn.Module) so the conv is registered once, under its reference key."""

  def __init__(self, conv, norm):
    self.conv, self.norm = conv, norm

  def fwd(self, x):
    fused = self.conv.fwd_moments(x)
    if fused is not None:                      # the norm's moment pass ran in the conv epilogue
      h, c1, mom, slots = fused
      y, c2 = self.norm.fwd_from_moments(h, mom, slots)
      return y, Ctx(c1, c2)
    h, c1 = self.conv.fwd(x)
    y, c2 = self.norm.fwd(h)
    return y, Ctx(c1, c2)

  def bwd(self, ctx, dy, need_dx=True, need_dw=True, addend=None, **conv_kw):
    c1, c2 = ctx.items
    dh = self.norm.bwd(c2, dy)
    return self.conv.bwd(c1, dh, need_dx, need_dw, addend=addend, **conv_kw)

  def dy_sink(self, ctx):
    """The norm that consumes the gradient handed to bwd(): run_chain_bwd passes it to the stage that produces that gradient."""
    return self.norm.sink(ctx.items[1])


class _Slot(nn.Module):
  """Parameter-free placeholder keeping nn.Sequential indices equal to the reference's."""

  def forward(self, x):
    return x


class HipResnetBlock(nn.Module):
  """x + IN(conv3(reflpad(ReLU(IN(conv3(reflpad(x)))))))   (networks.py:266-305)."""

  def __init__(self, dim, dtype=F32, device=None):
    super(HipResnetBlock, self).__init__()
    mk = lambda: HipConv2d(dim, dim, 3, 1, 1, PAD_REFLECT, apply_bias=False, dtype=dtype, device=device)
    self.conv_block = nn.Sequential(_Slot(), mk(), _Slot(), _Slot(), _Slot(), mk(), _Slot())
    self.norm1 = InstNormAct(ACT_RELU)
    self.norm2 = InstNormAct(ACT_NONE)

  # Activation checkpointing (BASELINE config 5): keep only the block's input and run the forward again
  # inside backward.  Off by default -- 288 GB of HBM hold every activation of the 2048x1024 step -- and, the
  # kernels being deterministic, bit-identical to the stored-activation path when on (tests/test_hip_configs.py).
  recompute = False

  @staticmethod
  def _conv_norm(conv, norm, x, residual=None):
    """conv -> InstanceNorm (+ activation / residual); the norm's moment pass runs in the conv's epilogue when its kernel has
    one (jpdse_conv_fwd_moments: the halo kernel of the full-width blocks), leaving ONE kernel per norm."""
    fused = conv.fwd_moments(x)
    if fused is not None:
      h, c, mom, slots = fused
      y, n = norm.fwd_from_moments(h, mom, slots, residual=residual)
      return y, c, n
    h, c = conv.fwd(x)
    y, n = norm.fwd(h, residual=residual)
    return y, c, n

  def _fwd(self, x):
    h, c1, n1 = self._conv_norm(self.conv_block[1], self.norm1, x)
    y, c2, n2 = self._conv_norm(self.conv_block[5], self.norm2, h, residual=x)
    return y, Ctx(c1, n1, c2, n2)

  def fwd(self, x):
    y, ctx = self._fwd(x)
    return (y, Ctx(x)) if self.recompute else (y, ctx)

  accepts_dx_sink = True

  def bwd(self, ctx, dy, need_dx=True, need_dw=True, dx_sink=None):
    """dx_sink: the InstanceNorm that consumes the returned gradient (the previous stage's, see dy_sink): the sums of its
    backward are then written by the epilogue of this block's first conv's data gradient; likewise norm1's by the second conv's
    -- no separate pass over (x, dy) for either (jpdse_conv_dgrad_fused_nsums)."""
    ctx = self.materialize(ctx)
    c1, n1, c2, n2 = ctx.items
    d = self.norm2.bwd(n2, dy)
    d = self.conv_block[5].bwd(c2, d, True, need_dw, sink=self.norm1.sink(n1) if self.fuse_norm_sums else None)
    d = self.norm1.bwd(n1, d)
    # the skip connection's gradient is summed in the data-gradient epilogue of the first conv
    return self.conv_block[1].bwd(c1, d, need_dx, need_dw, addend=dy if need_dx else None,
                                  sink=dx_sink if self.fuse_norm_sums else None)

  fuse_norm_sums = True       # False (A/B, tests): every norm backward computes its sums in its own pass

  def materialize(self, ctx):
    """The block's saved tensors: `ctx` itself, or -- checkpointed block (ctx holds only the block input) -- rebuilt by running
    the block's forward again.  run_chain_bwd calls this one stage early so that the stage above can be told about this block's
    second norm (dy_sink) in the checkpointed run exactly as in the stored-activation run: same kernels, bit-identical step."""
    if len(ctx.items) == 1:
      _, ctx = self._fwd(ctx.items[0])
    return ctx

  def dy_sink(self, ctx):
    if len(ctx.items) == 1:                     # checkpointed and not materialized
      return None
    return self.norm2.sink(ctx.items[3])


def run_chain_fwd(stages, x):
  ctxs = []
  for st in stages:
    x, c = st.fwd(x)
    ctxs.append(c)
  return x, ctxs


def run_chain_bwd(stages, ctxs, dy, need_dx=True, need_dw=True):
  """Back-propagate through `stages`; the first stage computes dx only when need_dx."""
  pending = None          # stage i - 1's saved tensors, materialized while stage i runs (the caller's list is left alone)
  for i in range(len(stages) - 1, -1, -1):
    ctx_i, pending = (pending if pending is not None else ctxs[i]), None
    below = None
    if i > 0 and hasattr(stages[i - 1], 'materialize'):
      below = pending = stages[i - 1].materialize(ctxs[i - 1])   # a checkpointed block's tensors, one stage early (see there)
    elif i > 0:
      below = ctxs[i - 1]
    if i > 0 and getattr(stages[i], 'accepts_dx_sink', False) and hasattr(stages[i - 1], 'dy_sink'):
      # stage i's dx goes straight into the InstanceNorm backward of stage i - 1: let its data-gradient epilogue write that
      # norm's sums (HipResnetBlock.bwd)
      dy = stages[i].bwd(ctx_i, dy, True, need_dw, dx_sink=stages[i - 1].dy_sink(below))
    else:
      dy = stages[i].bwd(ctx_i, dy, need_dx or i > 0, need_dw)
  return dy
