"""Functional torch-CPU restatement of the reference networks (TEST INFRASTRUCTURE).

Every network is a pure function of a state dict whose keys and shapes are the
reference checkpoint layout (SURVEY.md §8b "Checkpoint format"), so weights move
freely between the reference, this oracle and the HIP product.

Reference sites restated here (paths relative to /root/reference):
  * GlobalGenerator          ctu/models/pix2pixHD_networks/networks.py:198-251
  * ResnetBlock              networks.py:266-305
  * LocalEnhancer            networks.py:144-196
  * MultiscaleDiscriminator  networks.py:371-419  (getIntermFeat=True branch)
  * NLayerDiscriminator      networks.py:422-461
  * Vgg19 / VGGLoss          networks.py:474-504, 124-139
  * GANLoss (LSGAN)          networks.py:80-122
  * weights_init / define_*  networks.py:19-66
"""
from collections import OrderedDict
import math

import torch
import torch.nn.functional as F

IN_EPS = 1e-5          # nn.InstanceNorm2d default eps, affine=False (networks.py:31)
LRELU_SLOPE = 0.2      # networks.py:430,438,446
VGG_CFG = (64, 64, 'M', 128, 128, 'M', 256, 256, 256, 256, 'M',
           512, 512, 512, 512, 'M', 512)  # torchvision vgg19 features[0:30]
VGG_TAPS = (0, 2, 4, 8, 12)  # conv ordinal after whose ReLU a slice ends (relu1_1..relu5_1)
VGG_LOSS_WEIGHTS = (1.0 / 32, 1.0 / 16, 1.0 / 8, 1.0 / 4, 1.0)  # networks.py:132


# --------------------------------------------------------------------------- #
# optional bf16-STORAGE emulation (no reference counterpart: SURVEY.md §2.2)
# --------------------------------------------------------------------------- #
# The product's bf16 mode stores every activation AND every activation gradient in bf16 (fp32
# accumulation inside a kernel, fp32 weight gradients).  With `storage_bf16(True)` the oracle rounds at
# the same points -- the output of every conv (pre-norm), of every norm+activation(+residual), of every
# pool, and the gradients flowing back through those points -- and rounds the weights each conv sees,
# so the GPU bf16 path has a like-for-like CPU yardstick instead of only a loose bound against fp32.
class _RoundBF16(torch.autograd.Function):
  @staticmethod
  def forward(ctx, x):
    return x.to(torch.bfloat16).to(x.dtype)

  @staticmethod
  def backward(ctx, g):
    return g.to(torch.bfloat16).to(g.dtype)


_STORAGE_BF16 = [False]


def storage_bf16(on):
  _STORAGE_BF16[0] = bool(on)


def q(x):
  """Round an activation (and, in backward, its gradient) to bf16 storage when emulation is on."""
  return _RoundBF16.apply(x) if _STORAGE_BF16[0] else x


def qw(w):
  """Weights as the packed bf16 GEMM panels see them; the fp32 master keeps the full gradient."""
  if not _STORAGE_BF16[0]:
    return w
  return w + (w.detach().to(torch.bfloat16).to(w.dtype) - w.detach())


# --------------------------------------------------------------------------- #
# primitive layers
# --------------------------------------------------------------------------- #
def inorm(x):
  """InstanceNorm2d(affine=False, track_running_stats=False): biased variance."""
  return F.instance_norm(x, eps=IN_EPS)


def conv_reflect(x, w, b, pad):
  return q(F.conv2d(F.pad(x, (pad, pad, pad, pad), mode='reflect'), qw(w), b))


def resblock(sd, prefix, x):
  """x + IN(conv3(reflpad(ReLU(IN(conv3(reflpad(x)))))))   networks.py:271-305"""
  h = conv_reflect(x, sd[prefix + '.conv_block.1.weight'], sd[prefix + '.conv_block.1.bias'], 1)
  h = q(F.relu(inorm(h)))
  h = conv_reflect(h, sd[prefix + '.conv_block.5.weight'], sd[prefix + '.conv_block.5.bias'], 1)
  return q(x + inorm(h))


# --------------------------------------------------------------------------- #
# key layout helpers
# --------------------------------------------------------------------------- #
def global_layout(n_down, n_blocks):
  """Sequential indices of the parametrised modules of GlobalGenerator.model."""
  first = 1
  down = [4 + 3 * i for i in range(n_down)]
  res = [4 + 3 * n_down + b for b in range(n_blocks)]
  up = [4 + 3 * n_down + n_blocks + 3 * i for i in range(n_down)]
  last = 4 + 3 * n_down + n_blocks + 3 * n_down + 1
  return first, down, res, up, last


# --------------------------------------------------------------------------- #
# generators
# --------------------------------------------------------------------------- #
def global_trunk(sd, x, n_down, n_blocks, prefix='model'):
  """GlobalGenerator without its last [ReflPad3, Conv7, Tanh] (networks.py:153)."""
  first, down, res, up, _ = global_layout(n_down, n_blocks)
  p = lambda i, s: '%s.%d.%s' % (prefix, i, s)
  h = q(F.relu(inorm(conv_reflect(x, sd[p(first, 'weight')], sd[p(first, 'bias')], 3))))
  for i in down:
    h = q(F.relu(inorm(q(F.conv2d(h, qw(sd[p(i, 'weight')]), sd[p(i, 'bias')], stride=2, padding=1)))))
  for i in res:
    h = resblock(sd, '%s.%d' % (prefix, i), h)
  for i in up:
    h = q(F.conv_transpose2d(h, qw(sd[p(i, 'weight')]), sd[p(i, 'bias')],
                             stride=2, padding=1, output_padding=1))
    h = q(F.relu(inorm(h)))
  return h


def global_generator(sd, x, n_down=4, n_blocks=9):
  """networks.py:249-251 (mode='get_continuous_img')."""
  last = global_layout(n_down, n_blocks)[4]
  h = global_trunk(sd, x, n_down, n_blocks)
  h = F.conv2d(F.pad(h, (3, 3, 3, 3), mode='reflect'), qw(sd['model.%d.weight' % last]), sd['model.%d.bias' % last])
  return q(torch.tanh(h))     # bias + tanh are fused in the conv epilogue: one rounding


def avgpool3s2(x):
  """nn.AvgPool2d(3, stride=2, padding=[1,1], count_include_pad=False)  networks.py:180,387"""
  return q(F.avg_pool2d(x, 3, stride=2, padding=1, count_include_pad=False))


def local_enhancer(sd, x, n_down=4, n_blocks=9, n_local=1, n_blocks_local=3):
  """networks.py:182-196."""
  pyramid = [x]
  for _ in range(n_local):
    pyramid.append(avgpool3s2(pyramid[-1]))
  out = global_trunk(sd, pyramid[-1], n_down, n_blocks)
  for n in range(1, n_local + 1):
    d, u = 'model%d_1' % n, 'model%d_2' % n
    xi = pyramid[n_local - n]
    h = q(F.relu(inorm(conv_reflect(xi, sd[d + '.1.weight'], sd[d + '.1.bias'], 3))))
    h = q(F.relu(inorm(q(F.conv2d(h, qw(sd[d + '.4.weight']), sd[d + '.4.bias'], stride=2, padding=1)))))
    h = q(h + out)
    for b in range(n_blocks_local):
      h = resblock(sd, '%s.%d' % (u, b), h)
    k = n_blocks_local
    h = q(F.conv_transpose2d(h, qw(sd['%s.%d.weight' % (u, k)]), sd['%s.%d.bias' % (u, k)],
                             stride=2, padding=1, output_padding=1))
    h = q(F.relu(inorm(h)))
    out = h
    if n == n_local:
      out = q(torch.tanh(F.conv2d(F.pad(h, (3, 3, 3, 3), mode='reflect'),
                                  qw(sd['%s.%d.weight' % (u, k + 4)]), sd['%s.%d.bias' % (u, k + 4)])))
  return out


def generator(sd, x, cfg):
  if cfg['netG'] == 'global':
    return global_generator(sd, x, cfg['n_downsample_global'], cfg['n_blocks_global'])
  if cfg['netG'] == 'local':
    return local_enhancer(sd, x, cfg['n_downsample_global'], cfg['n_blocks_global'],
                          cfg['n_local_enhancers'], cfg['n_blocks_local'])
  raise ValueError('generator not implemented: %r' % (cfg['netG'],))


# --------------------------------------------------------------------------- #
# discriminator
# --------------------------------------------------------------------------- #
def nlayer_d(sd, prefix, x, n_layers=3):
  """One PatchGAN scale, all intermediate features (networks.py:389-397, 430-449)."""
  feats = []
  h = x
  for j in range(n_layers + 2):
    w, b = sd['%s_layer%d.0.weight' % (prefix, j)], sd['%s_layer%d.0.bias' % (prefix, j)]
    stride = 2 if j < n_layers else 1
    h = F.conv2d(h, qw(w), b, stride=stride, padding=2)
    if 0 < j <= n_layers:
      h = inorm(q(h))
    if j <= n_layers:
      h = F.leaky_relu(h, LRELU_SLOPE)
    h = q(h)
    feats.append(h)
  return feats


def multiscale_d(sd, x, num_D=2, n_layers=3):
  """networks.py:404-419: scale num_D-1-i sees the i-times pooled input."""
  out, h = [], x
  for i in range(num_D):
    out.append(nlayer_d(sd, 'scale%d' % (num_D - 1 - i), h, n_layers))
    if i != num_D - 1:
      h = avgpool3s2(h)
  return out


# --------------------------------------------------------------------------- #
# VGG19 feature pyramid
# --------------------------------------------------------------------------- #
def vgg19_features(sd, x):
  """features[0:30] split after relu1_1,2_1,3_1,4_1,5_1 (networks.py:483-504).
  Inputs are NOT ImageNet-normalised in the reference."""
  maps, h, ci = [], x, 0
  for item in VGG_CFG:
    if item == 'M':
      h = F.max_pool2d(h, 2, 2)
      continue
    h = q(F.relu(F.conv2d(h, qw(sd['vgg.%d.weight' % ci]), sd['vgg.%d.bias' % ci], padding=1)))
    if ci in VGG_TAPS:
      maps.append(h)
    ci += 1
  return maps


# --------------------------------------------------------------------------- #
# losses
# --------------------------------------------------------------------------- #
def gan_loss(preds, target_is_real):
  """LSGAN: sum over scales of mean((last_feature - t)^2)  (networks.py:112-119)."""
  t = 1.0 if target_is_real else 0.0
  total = 0
  for scale in preds:
    p = scale[-1]
    total = total + F.mse_loss(p, torch.full_like(p, t))
  return total


def vgg_loss(sd_vgg, fake, real):
  fx, fy = vgg19_features(sd_vgg, fake), vgg19_features(sd_vgg, real)
  total = 0
  for w, a, b in zip(VGG_LOSS_WEIGHTS, fx, fy):
    total = total + w * F.l1_loss(a, b.detach())
  return total


# --------------------------------------------------------------------------- #
# parameter construction in the reference's RNG order
# --------------------------------------------------------------------------- #
def _conv_params(cin, cout, k, transposed=False):
  """Instantiate the torch module only to consume the RNG exactly as the
  reference's constructor does (kaiming-uniform weight, uniform bias)."""
  m = (torch.nn.ConvTranspose2d(cin, cout, k, stride=2, padding=1, output_padding=1)
       if transposed else torch.nn.Conv2d(cin, cout, k))
  return m.weight.detach(), m.bias.detach()


def _finish(sd, std=0.02):
  """define_G/define_D end with net.apply(weights_init): every *Conv* weight is
  redrawn N(0, 0.02) in module-traversal order; biases keep their constructor
  values (networks.py:19-25,55,65)."""
  for k, v in sd.items():
    if k.endswith('.weight'):
      v.normal_(0.0, std)
  return sd


def _global_params(sd, prefix, input_nc, output_nc, ngf, n_down, n_blocks, keep_last=True):
  first, down, res, up, last = global_layout(n_down, n_blocks)
  def put(idx, wb, sub=''):
    sd['%s.%d%s.weight' % (prefix, idx, sub)], sd['%s.%d%s.bias' % (prefix, idx, sub)] = wb
  put(first, _conv_params(input_nc, ngf, 7))
  for i, idx in enumerate(down):
    put(idx, _conv_params(ngf * 2 ** i, ngf * 2 ** (i + 1), 3))
  dim = ngf * 2 ** n_down
  for idx in res:
    put(idx, _conv_params(dim, dim, 3), '.conv_block.1')
    put(idx, _conv_params(dim, dim, 3), '.conv_block.5')
  for i, idx in enumerate(up):
    c = ngf * 2 ** (n_down - i)
    put(idx, _conv_params(c, c // 2, 3, transposed=True))
  wb = _conv_params(ngf, output_nc, 7)   # always constructed -> always consumes RNG
  if keep_last:
    put(last, wb)


def init_generator(cfg, input_nc, output_nc=3):
  """State dict equal to define_G(...).state_dict() under the current torch RNG."""
  sd = OrderedDict()
  ngf, nd, nb = cfg['ngf'], cfg['n_downsample_global'], cfg['n_blocks_global']
  if cfg['netG'] == 'global':
    _global_params(sd, 'model', input_nc, output_nc, ngf, nd, nb)
    return _finish(sd)
  if cfg['netG'] != 'local':
    raise ValueError('generator not implemented: %r' % (cfg['netG'],))
  nl, nbl = cfg['n_local_enhancers'], cfg['n_blocks_local']
  _global_params(sd, 'model', input_nc, output_nc, ngf * 2 ** nl, nd, nb, keep_last=False)
  for n in range(1, nl + 1):
    g = ngf * 2 ** (nl - n)
    d, u = 'model%d_1' % n, 'model%d_2' % n
    down = OrderedDict()
    down[d + '.1.weight'], down[d + '.1.bias'] = _conv_params(input_nc, g, 7)
    down[d + '.4.weight'], down[d + '.4.bias'] = _conv_params(g, 2 * g, 3)
    upd = OrderedDict()
    for b in range(nbl):
      for sub in ('1', '5'):
        w, bb = _conv_params(2 * g, 2 * g, 3)
        upd['%s.%d.conv_block.%s.weight' % (u, b, sub)] = w
        upd['%s.%d.conv_block.%s.bias' % (u, b, sub)] = bb
    upd['%s.%d.weight' % (u, nbl)], upd['%s.%d.bias' % (u, nbl)] = \
        _conv_params(2 * g, g, 3, transposed=True)
    if n == nl:
      upd['%s.%d.weight' % (u, nbl + 4)], upd['%s.%d.bias' % (u, nbl + 4)] = \
          _conv_params(ngf, output_nc, 7)
    sd.update(down)
    sd.update(upd)
  return _finish(sd)


def init_discriminator(input_nc, ndf=64, n_layers=3, num_D=2):
  """State dict equal to define_D(..., getIntermFeat=True).state_dict()."""
  sd = OrderedDict()
  for i in range(num_D):
    chans = [input_nc, ndf]
    nf = ndf
    for _ in range(1, n_layers):
      nf = min(nf * 2, 512)
      chans.append(nf)
    chans.append(min(nf * 2, 512))
    chans.append(1)
    for j in range(n_layers + 2):
      w, b = _conv_params(chans[j], chans[j + 1], 4)
      sd['scale%d_layer%d.0.weight' % (i, j)] = w
      sd['scale%d_layer%d.0.bias' % (i, j)] = b
  return _finish(sd)


def init_vgg19(seed=20):
  """Seeded He-normal VGG19 'E' feature weights (the ImageNet checkpoint is a
  download that is unavailable offline: SURVEY.md §8c).  Keys 'vgg.<ordinal>'."""
  g = torch.Generator().manual_seed(seed)
  sd, cin, ci = OrderedDict(), 3, 0
  for item in VGG_CFG:
    if item == 'M':
      continue
    std = math.sqrt(2.0 / (cin * 9))
    sd['vgg.%d.weight' % ci] = torch.randn(item, cin, 3, 3, generator=g) * std
    sd['vgg.%d.bias' % ci] = torch.randn(item, generator=g) * 0.01
    cin, ci = item, ci + 1
  return sd


def vgg_torchvision_keys(sd):
  """Re-key 'vgg.<ordinal>' to torchvision's `features.<idx>` numbering."""
  out, idx, ci = OrderedDict(), 0, 0
  for item in VGG_CFG:
    if item == 'M':
      idx += 1
      continue
    out['features.%d.weight' % idx] = sd['vgg.%d.weight' % ci]
    out['features.%d.bias' % idx] = sd['vgg.%d.bias' % ci]
    idx, ci = idx + 2, ci + 1
  return out
