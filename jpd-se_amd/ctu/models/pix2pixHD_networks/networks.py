"""MI355X-native pix2pixHD networks behind the reference's `networks` interface.

Drop-in for /root/reference/ctu/models/pix2pixHD_networks/networks.py: same factory
signatures (`define_G` :38-56, `define_D` :58-66), same class names, same `state_dict`
keys/shapes.  Every layer runs as a hand-written gfx950 kernel through
libjpdse_hip.so; there is no torch.nn compute and no CPU fallback.

Two entry styles per network:
  * `net(x)` -- reference style: fp32 NCHW cuda tensor in, fp32 NCHW tensor(s) out (inference);
  * `net.fwd(act)` / `net.bwd(...)` -- NHWC `Act` in/out with an explicit backward, used by
    `Pix2PixHDModel`'s train-step schedule.
"""
import torch
import torch.nn as nn

from jpdse_hip import (ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH, PAD_ZERO, PAD_REFLECT, F32, BF16, JpdseError,
                       require_gpu)
from jpdse_hip import ops
from jpdse_hip.ops import Act
from jpdse_hip.layers import (HipConv2d, HipResnetBlock, InstNormAct, ConvNormAct, Ctx, _Slot, run_chain_fwd,
                              run_chain_bwd)

ResnetBlock = HipResnetBlock

VGG_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M', 512, 512, 512, 512, 'M', 512)
VGG_TAPS = (0, 2, 4, 8, 12)


def dtype_code(name):
  if name in (F32, BF16):
    return name
  table = {'fp32': F32, 'float32': F32, 'f32': F32, 'bf16': BF16, 'bfloat16': BF16}
  if name not in table:
    raise ValueError('compute dtype must be fp32 or bf16, got %r' % (name,))
  return table[name]


def get_norm_layer(norm_type='instance'):
  """Only the reference default is on the accelerated path (networks.py:27-36)."""
  if norm_type != 'instance':
    raise NotImplementedError('normalization layer [%s] is not implemented on the HIP path '
                              '(per-image data parallelism relies on InstanceNorm: SURVEY.md §8e)' % norm_type)
  return InstNormAct


def weights_init(m):
  """N(0, 0.02) on every conv weight (networks.py:19-25)."""
  if isinstance(m, HipConv2d):
    m.weight.data.normal_(0.0, 0.02)


def _to_act(x, dtype):
  if isinstance(x, Act):
    return x
  if not (torch.is_tensor(x) and x.is_cuda):
    raise JpdseError('HIP networks take cuda tensors (no CPU fallback); got %r' % (type(x),))
  return ops.nchw_to_nhwc(x.float().contiguous(), dtype)


# =============================================================================================
# generators
# =============================================================================================
def _global_indices(n_down, n_blocks):
  first = 1
  down = [4 + 3 * i for i in range(n_down)]
  res = [4 + 3 * n_down + b for b in range(n_blocks)]
  up = [4 + 3 * n_down + n_blocks + 3 * i for i in range(n_down)]
  last = 4 + 3 * n_down + n_blocks + 3 * n_down + 1
  return first, down, res, up, last


def _build_global(input_nc, output_nc, ngf, n_down, n_blocks, dtype, device, with_head=True):
  """Returns (nn.Sequential with reference indices, trunk stage list, head stage or None)."""
  first, down, res, up, last = _global_indices(n_down, n_blocks)
  total = last + 2 if with_head else last - 1     # drop [ReflPad, Conv7, Tanh] for the enhancer's core
  slots = [_Slot() for _ in range(total)]
  kw = dict(dtype=dtype, device=device)
  stages = []
  slots[first] = HipConv2d(input_nc, ngf, 7, 1, 3, PAD_REFLECT, apply_bias=False, **kw)
  stages.append(ConvNormAct(slots[first], InstNormAct(ACT_RELU)))
  for i, idx in enumerate(down):
    c = ngf * 2 ** i
    slots[idx] = HipConv2d(c, 2 * c, 3, 2, 1, PAD_ZERO, apply_bias=False, **kw)
    stages.append(ConvNormAct(slots[idx], InstNormAct(ACT_RELU)))
  dim = ngf * 2 ** n_down
  for idx in res:
    slots[idx] = HipResnetBlock(dim, **kw)
    stages.append(slots[idx])
  for i, idx in enumerate(up):
    c = ngf * 2 ** (n_down - i)
    slots[idx] = HipConv2d(c, c // 2, 3, 2, 1, transposed=True, apply_bias=False, **kw)
    stages.append(ConvNormAct(slots[idx], InstNormAct(ACT_RELU)))
  head = None
  if with_head:
    slots[last] = HipConv2d(ngf, output_nc, 7, 1, 3, PAD_REFLECT, act=ACT_TANH, apply_bias=True, **kw)
    head = slots[last]
  return nn.Sequential(*slots), stages, head


class GlobalGenerator(nn.Module):
  """networks.py:198-251 (binarizer branches are outside the JPD-SE flag subset)."""

  def __init__(self, input_nc, output_nc, ngf=64, n_downsampling=3, n_blocks=9, norm_layer=None,
               padding_type='reflect', binarize=False, binarizer_out_channels=128, bin_before_res=True,
               compute_dtype='fp32', device=None):
    super(GlobalGenerator, self).__init__()
    assert n_blocks >= 0
    if binarize:
      raise NotImplementedError('generator binarization is outside the JPD-SE hot path (SURVEY.md §2 row 3)')
    if padding_type != 'reflect':
      raise NotImplementedError('padding [%s] is not implemented' % padding_type)
    self.binarize = False
    self.n_downsampling, self.n_blocks = n_downsampling, n_blocks
    self.input_nc, self.output_nc = input_nc, output_nc
    self.cdtype = dtype_code(compute_dtype)
    self.model, stages, head = _build_global(input_nc, output_nc, ngf, n_downsampling, n_blocks, self.cdtype,
                                             device)
    self._stages = stages + [head]

  def fwd(self, x):
    return run_chain_fwd(self._stages, x)

  def bwd(self, ctxs, dy, need_dx=False, need_dw=True):
    return run_chain_bwd(self._stages, ctxs, dy, need_dx, need_dw)

  def forward(self, input, mode='get_continuous_img'):
    if mode == 'get_binary_code':
      raise AttributeError('Generator: no binarizer found')
    if mode != 'get_continuous_img':
      raise ValueError('Invalid generator mode: {}'.format(mode))
    y, _ = self.fwd(_to_act(input, self.cdtype))
    return ops.nhwc_to_nchw(y)


class LocalEnhancer(nn.Module):
  """networks.py:144-196: coarse GlobalGenerator(ngf*2^n) core on the pooled input, then per
  enhancer [ReflPad3,Conv7,IN,ReLU,Conv3s2,IN,ReLU](full-res) + coarse -> ResBlocks -> ConvT."""

  def __init__(self, input_nc, output_nc, ngf=32, n_downsample_global=3, n_blocks_global=9, n_local_enhancers=1,
               n_blocks_local=3, norm_layer=None, padding_type='reflect', compute_dtype='fp32', device=None):
    super(LocalEnhancer, self).__init__()
    if padding_type != 'reflect':
      raise NotImplementedError('padding [%s] is not implemented' % padding_type)
    self.n_local_enhancers = n_local_enhancers
    self.input_nc, self.output_nc = input_nc, output_nc
    self.cdtype = dtype_code(compute_dtype)
    kw = dict(dtype=self.cdtype, device=device)
    self.model, self._core, _ = _build_global(input_nc, output_nc, ngf * 2 ** n_local_enhancers,
                                              n_downsample_global, n_blocks_global, self.cdtype, device,
                                              with_head=False)
    self._down, self._up = {}, {}
    for n in range(1, n_local_enhancers + 1):
      g = ngf * 2 ** (n_local_enhancers - n)
      c7 = HipConv2d(input_nc, g, 7, 1, 3, PAD_REFLECT, apply_bias=False, **kw)
      c3 = HipConv2d(g, 2 * g, 3, 2, 1, PAD_ZERO, apply_bias=False, **kw)
      setattr(self, 'model%d_1' % n, nn.Sequential(_Slot(), c7, _Slot(), _Slot(), c3, _Slot(), _Slot()))
      self._down[n] = [ConvNormAct(c7, InstNormAct(ACT_RELU)), ConvNormAct(c3, InstNormAct(ACT_RELU))]
      blocks = [HipResnetBlock(2 * g, **kw) for _ in range(n_blocks_local)]
      ct = HipConv2d(2 * g, g, 3, 2, 1, transposed=True, apply_bias=False, **kw)
      mods = blocks + [ct, _Slot(), _Slot()]
      stages = blocks + [ConvNormAct(ct, InstNormAct(ACT_RELU))]
      if n == n_local_enhancers:
        head = HipConv2d(ngf, output_nc, 7, 1, 3, PAD_REFLECT, act=ACT_TANH, apply_bias=True, **kw)
        mods += [_Slot(), head, _Slot()]
        stages.append(head)
      setattr(self, 'model%d_2' % n, nn.Sequential(*mods))
      self._up[n] = stages

  def fwd(self, x):
    nl = self.n_local_enhancers
    pyramid = [x]
    for _ in range(nl):
      pyramid.append(ops.avgpool3s2_fwd(pyramid[-1]))
    out, core_ctx = run_chain_fwd(self._core, pyramid[-1])
    levels = []
    for n in range(1, nl + 1):
      h, dctx = run_chain_fwd(self._down[n], pyramid[nl - n])
      h = ops.add_(h, out)
      out, uctx = run_chain_fwd(self._up[n], h)
      levels.append((dctx, uctx))
    return out, (core_ctx, levels)

  def bwd(self, ctxs, dy, need_dx=False, need_dw=True):
    if need_dx:
      raise NotImplementedError('gradient w.r.t. the generator input is never needed on this path')
    core_ctx, levels = ctxs
    for n in range(self.n_local_enhancers, 0, -1):
      dctx, uctx = levels[n - 1]
      dh = run_chain_bwd(self._up[n], uctx, dy, True, need_dw)
      run_chain_bwd(self._down[n], dctx, dh, False, need_dw)
      dy = dh                       # the sum feeds both branches
    if any(p.requires_grad for p in self.model.parameters()):     # frozen during the niter_fix_global phase
      run_chain_bwd(self._core, core_ctx, dy, False, need_dw)
    return None

  def forward(self, input):
    y, _ = self.fwd(_to_act(input, self.cdtype))
    return ops.nhwc_to_nchw(y)


def define_G(input_nc, output_nc, ngf, netG, n_downsample_global=3, n_blocks_global=9, n_local_enhancers=1,
             n_blocks_local=3, norm='instance', gpu_ids=[], binarize_encoder=False,
             encoder_binarizer_out_channels=128, encoder_groups=1, binarize_generator=False,
             bin_generator_before_res=True, generator_binarizer_out_channels=128, compute_dtype='fp32'):
  get_norm_layer(norm)
  device = torch.device('cuda', gpu_ids[0]) if len(gpu_ids) > 0 else None
  if device is not None:
    require_gpu(gpu_ids[0])
  if netG == 'global':
    net = GlobalGenerator(input_nc, output_nc, ngf, n_downsample_global, n_blocks_global,
                          binarize=binarize_generator, compute_dtype=compute_dtype, device=device)
  elif netG == 'local':
    net = LocalEnhancer(input_nc, output_nc, ngf, n_downsample_global, n_blocks_global, n_local_enhancers,
                        n_blocks_local, compute_dtype=compute_dtype, device=device)
  elif netG == 'encoder':
    raise NotImplementedError('the learned-codec Encoder is outside the JPD-SE hot path (SURVEY.md §2 row 3)')
  else:
    raise ValueError('generator not implemented!')
  return net


# =============================================================================================
# discriminator
# =============================================================================================
class NLayerDiscriminator(nn.Module):
  """One PatchGAN scale with every intermediate feature returned (networks.py:422-471)."""

  def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=None, use_sigmoid=False, getIntermFeat=True,
               compute_dtype='fp32', device=None):
    super(NLayerDiscriminator, self).__init__()
    if use_sigmoid:
      raise NotImplementedError('--no_lsgan (sigmoid discriminator) is outside the JPD-SE hot path')
    self.n_layers = n_layers
    self.fuse_lrelu0 = True     # False: stage 0's LeakyReLU backward as its own pass (tests compare the two forms)
    self.cdtype = dtype_code(compute_dtype)
    kw = dict(dtype=self.cdtype, device=device)
    chans = [input_nc, ndf]
    nf = ndf
    for _ in range(1, n_layers):
      nf = min(nf * 2, 512)
      chans.append(nf)
    chans.append(min(nf * 2, 512))
    chans.append(1)
    self._stages = []
    for j in range(n_layers + 2):
      stride = 2 if j < n_layers else 1
      if j == 0:
        conv = HipConv2d(chans[0], chans[1], 4, stride, 2, PAD_ZERO, act=ACT_LRELU, slope=0.2, **kw)
        seq, stage = nn.Sequential(conv, _Slot()), conv
      elif j <= n_layers:
        conv = HipConv2d(chans[j], chans[j + 1], 4, stride, 2, PAD_ZERO, apply_bias=False, **kw)
        seq, stage = nn.Sequential(conv, _Slot(), _Slot()), ConvNormAct(conv, InstNormAct(ACT_LRELU, 0.2))
      else:
        conv = HipConv2d(chans[j], 1, 4, 1, 2, PAD_ZERO, **kw)
        seq, stage = nn.Sequential(conv), conv
      setattr(self, 'model' + str(j), seq)
      self._stages.append(stage)

  def fwd(self, x):
    feats, ctxs = [], []
    for st in self._stages:
      x, c = st.fwd(x)
      feats.append(x)
      ctxs.append(c)
    return feats, ctxs

  def bwd(self, ctxs, dfeats, need_dx=False, need_dw=True, dx_channels=None):
    """dfeats[j]: gradient w.r.t. feature j (Act) or None.  dx_channels=(c0, c1): return the input gradient of
    those channels only (needs need_dw False: the generator's pass through D)."""
    d = None
    fused = False           # dfeats[j] already summed into d by the data-gradient epilogue of stage j+1
    dz0 = False             # d is already the gradient w.r.t. stage 0's pre-activation
    for j in range(len(self._stages) - 1, -1, -1):
      if dfeats[j] is not None and not fused:
        d = dfeats[j] if d is None else ops.add_(d, dfeats[j])
      fused = False
      if d is None:
        continue
      # stage 1's input is stage 0's LeakyReLU(0.2) output: that activation's backward rides in the epilogue of
      # stage 1's data gradient (after the feature-matching addend), and stage 0 then starts from dz
      lrelu0 = dict(relu_input=True, input_slope=self._stages[0].slope) if (j == 1 and self.fuse_lrelu0) else {}
      if j == 0 and dx_channels is not None and need_dx and not need_dw:
        d = self._stages[0].bwd_input_slice(ctxs[0], d, dx_channels[0], dx_channels[1], dy_is_dz=dz0)
      elif j == 0:
        d = self._stages[0].bwd(ctxs[0], d, need_dx, need_dw, dy_is_dz=dz0)
      elif dfeats[j - 1] is not None:
        # the feature-matching gradient of feature j-1 joins in the epilogue of this stage's data gradient
        d = self._stages[j].bwd(ctxs[j], d, True, need_dw, addend=dfeats[j - 1], **lrelu0)
        fused = True
      else:
        d = self._stages[j].bwd(ctxs[j], d, True, need_dw, **lrelu0)
      dz0 = bool(lrelu0)
    return d


class MultiscaleDiscriminator(nn.Module):
  """networks.py:371-419 with getIntermFeat hard-wired True (pix2pixHD_model.py:162-163)."""

  def __init__(self, input_nc, ndf=64, n_layers=3, norm_layer=None, use_sigmoid=False, num_D=3,
               getIntermFeat=True, compute_dtype='fp32', device=None):
    super(MultiscaleDiscriminator, self).__init__()
    if not getIntermFeat:
      raise NotImplementedError('the JPD-SE model always requests intermediate features')
    self.num_D, self.n_layers, self.getIntermFeat = num_D, n_layers, True
    self.cdtype = dtype_code(compute_dtype)
    self._scales = []
    for i in range(num_D):
      netD = NLayerDiscriminator(input_nc, ndf, n_layers, norm_layer, use_sigmoid, True, compute_dtype, device)
      for j in range(n_layers + 2):
        setattr(self, 'scale' + str(i) + '_layer' + str(j), getattr(netD, 'model' + str(j)))
      self._scales.append(netD)     # plain list: parameters are registered through scale{i}_layer{j}

  def fwd(self, x):
    """Returns (result, ctx); result[i][j] is feature j of the i-times pooled input, computed by
    the sub-network stored as scale{num_D-1-i} (networks.py:408-418)."""
    result, ctxs, h = [], [], x
    for i in range(self.num_D):
      feats, c = self._scales[self.num_D - 1 - i].fwd(h)
      result.append(feats)
      ctxs.append((c, h.H, h.W))
      if i != self.num_D - 1:
        h = ops.avgpool3s2_fwd(h)
    return result, ctxs

  def bwd(self, ctxs, dresult, need_dx=False, need_dw=True, batch=None, dx_channels=None):
    """dresult[i][j]: Act or None.  `batch=(b0,b1)` back-propagates only that sub-batch of the
    saved forward (per-sample InstanceNorm makes sub-batches independent).  dx_channels=(c0, c1): the returned
    input gradient covers those input channels only (AvgPool between the scales acts per channel)."""
    dx = None
    for i in range(self.num_D - 1, -1, -1):
      c, H, W = ctxs[i]
      if batch is not None:
        c = [ci.slice(batch[0], batch[1]) for ci in c]
      d = self._scales[self.num_D - 1 - i].bwd(c, dresult[i], need_dx, need_dw, dx_channels)
      if need_dx:
        if dx is not None:          # gradient arriving from the coarser scale through AvgPool
          d = ops.add_(d, ops.avgpool3s2_bwd(dx, H, W))
        dx = d
    return dx

  def forward(self, input, keep_input=False):
    if keep_input:
      raise NotImplementedError('--match_raw_feat is outside the JPD-SE hot path')
    result, _ = self.fwd(_to_act(input, self.cdtype))
    return [[ops.nhwc_to_nchw(f) for f in scale] for scale in result]


def define_D(input_nc, ndf, n_layers_D, norm='instance', use_sigmoid=False, num_D=1, getIntermFeat=False,
             gpu_ids=[], compute_dtype='fp32'):
  get_norm_layer(norm)
  device = torch.device('cuda', gpu_ids[0]) if len(gpu_ids) > 0 else None
  if device is not None:
    require_gpu(gpu_ids[0])
  return MultiscaleDiscriminator(input_nc, ndf, n_layers_D, None, use_sigmoid, num_D, getIntermFeat,
                                 compute_dtype, device)


# =============================================================================================
# VGG19 feature pyramid + losses
# =============================================================================================
class Vgg19(nn.Module):
  """torchvision vgg19.features[0:30] cut after relu1_1, 2_1, 3_1, 4_1, 5_1 (networks.py:474-504).
  Frozen.  Weights: `load_torchvision_state_dict` (keys `features.<i>.weight/bias`) or, with no
  checkpoint available offline, the seeded He-normal initialisation the oracle also uses."""

  def __init__(self, requires_grad=False, compute_dtype='fp32', device=None, seed=20):
    super(Vgg19, self).__init__()
    self.cdtype = dtype_code(compute_dtype)
    convs, cin = [], 3
    g = torch.Generator().manual_seed(seed)
    for item in VGG_CFG:
      if item == 'M':
        continue
      conv = HipConv2d(cin, item, 3, 1, 1, PAD_ZERO, act=ACT_RELU, dtype=self.cdtype, device=device)
      std = (2.0 / (cin * 9)) ** 0.5
      with torch.no_grad():
        conv.weight.copy_(torch.randn(item, cin, 3, 3, generator=g) * std)
        conv.bias.copy_(torch.randn(item, generator=g) * 0.01)
      convs.append(conv)
      cin = item
    self.convs = nn.ModuleList(convs)
    for p in self.parameters():
      p.requires_grad = requires_grad

  def load_torchvision_state_dict(self, sd):
    idx, ci = 0, 0
    with torch.no_grad():
      for item in VGG_CFG:
        if item == 'M':
          idx += 1
          continue
        self.convs[ci].weight.copy_(sd['features.%d.weight' % idx])
        self.convs[ci].bias.copy_(sd['features.%d.bias' % idx])
        idx, ci = idx + 2, ci + 1

  def fwd(self, x, save=True):
    maps, ctxs, ci = [], [], 0
    pooled = None
    for pos, item in enumerate(VGG_CFG):
      if item == 'M':
        y = pooled if pooled is not None else ops.maxpool2_fwd(x)     # written by the conv in front of it (jpdse_conv_fwd_pool)
        pooled = None
        ctxs.append(Ctx(x) if save else None)
        x = y
        continue
      if pos + 1 < len(VGG_CFG) and VGG_CFG[pos + 1] == 'M' and x.H % 2 == 0 and x.W % 2 == 0:
        x, pooled, c = self.convs[ci].fwd_pool(x)
      else:
        x, c = self.convs[ci].fwd(x)
      ctxs.append(c if save else None)
      if ci in VGG_TAPS:
        maps.append(x)
      ci += 1
    return maps, ctxs

  def bwd(self, ctxs, dmaps):
    """Gradient w.r.t. the input image given the gradients of the five taps (dgrad only).
    Every ReLU backward is fused: `dmaps` must already be masked by their tap's ReLU
    (ops.l1_bwd(..., relu_a=True)), and each conv's data gradient is masked by its own input
    (a ReLU output or the max-pool of one) in the GEMM epilogue, so the gradient that reaches a
    conv is always w.r.t. its pre-activation."""
    order, ci = [], 0
    for item in VGG_CFG:
      order.append(('M', None) if item == 'M' else ('C', ci))
      if item != 'M':
        ci += 1
    d = None
    for pos in range(len(order) - 1, -1, -1):
      kind, ci = order[pos]
      if kind == 'C' and ci in VGG_TAPS and d is None:
        d = dmaps[VGG_TAPS.index(ci)]            # deepest tap: nothing to add to
      if d is None:
        continue
      if kind == 'M':
        (x,) = ctxs[pos].items
        d = ops.maxpool2_bwd(x, d)
      else:
        # the loss gradient of the tap that is this conv's INPUT (relu{k}_1 feeds conv{k}_2) is summed in the epilogue
        tap = dmaps[VGG_TAPS.index(ci - 1)] if (ci - 1) in VGG_TAPS and order[pos - 1][0] == 'C' else None
        d = self.convs[ci].bwd(ctxs[pos], d, True, False, dy_is_dz=True, relu_input=ci > 0, addend=tap)
    return d

  def forward(self, X):
    maps, _ = self.fwd(_to_act(X, self.cdtype), save=False)
    return [ops.nhwc_to_nchw(m) for m in maps]


class VGGLoss(nn.Module):
  """networks.py:124-139."""
  weights = [1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0]

  def __init__(self, gpu_ids, compute_dtype='fp32'):
    super(VGGLoss, self).__init__()
    device = torch.device('cuda', gpu_ids[0]) if len(gpu_ids) else None
    self.vgg = Vgg19(compute_dtype=compute_dtype, device=device)

  def forward(self, x, y):
    dt = self.vgg.cdtype
    fx, _ = self.vgg.fwd(_to_act(x, dt), save=False)
    fy, _ = self.vgg.fwd(_to_act(y, dt), save=False)
    slots = torch.zeros(len(fx), dtype=torch.float32, device=fx[0].t.device)
    for i in range(len(fx)):
      ops.l1_fwd(fx[i], fy[i], slots[i:i + 1])
    vals = slots.cpu().tolist()
    return torch.tensor(sum(w * v for w, v in zip(self.weights, vals)), device=fx[0].t.device)


class GANLoss(nn.Module):
  """LSGAN objective on the last feature of every scale (networks.py:80-122)."""

  def __init__(self, use_lsgan=True, target_real_label=1.0, target_fake_label=0.0, tensor=None):
    super(GANLoss, self).__init__()
    if not use_lsgan:
      raise NotImplementedError('--no_lsgan is outside the JPD-SE hot path')
    self.real_label, self.fake_label = target_real_label, target_fake_label

  def __call__(self, input, target_is_real):
    t = self.real_label if target_is_real else self.fake_label
    preds = [s[-1] for s in input] if isinstance(input[0], list) else [input[-1]]
    dev = preds[0].t.device if isinstance(preds[0], Act) else preds[0].device
    slots = torch.zeros(len(preds), dtype=torch.float32, device=dev)
    for i, p in enumerate(preds):
      a = p if isinstance(p, Act) else ops.nchw_to_nhwc(p.float().contiguous(), F32)
      ops.mse_const_fwd(a, t, slots[i:i + 1])
    return torch.tensor(sum(slots.cpu().tolist()), device=dev)
