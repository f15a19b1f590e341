"""torch-CPU restatement of the JPD-SE loss graph and train step (TEST INFRASTRUCTURE).

Reference sites (paths relative to /root/reference):
  * Pix2PixHDModel.preprocess / get_edges    ctu/models/pix2pixHD_model.py:362-412, 774-783
  * Pix2PixHDModel._get_img (JPD-SE branch)  pix2pixHD_model.py:508-518, 594-595, 608-610
  * Pix2PixHDModel.discriminate              pix2pixHD_model.py:451-460 (pool_size == 0)
  * Pix2PixHDModel.get_train_loss            pix2pixHD_model.py:709-771
  * Pix2PixHDModel.get_eval_loss             pix2pixHD_model.py:621-643
  * tensor2im                                ctu/utils/misc.py:64-95
  * Pix2PixHDTrainer.step                    ctu/trainers/pix2pixHD_trainer.py:42-85
  * create_optimizers                        pix2pixHD_model.py:248-280

Only the flag subset of scripts/pix2pixHD_bpg_train.sh is restated
(no_label_encoding, no_feat_encoding, no_generator_binarization, instance edges
on, LSGAN, pool_size 0); see SURVEY.md §2 rows 2-3.
"""
from collections import OrderedDict
import types

import numpy as np
import torch
import torch.nn.functional as F

from . import nets

LOSS_NAMES = ('G_GAN', 'G_GAN_Feat', 'G_VGG', 'G_Distortion', 'D_real', 'D_fake')


def default_opt(**over):
  """The `opt` fields the hot path reads, script defaults."""
  o = dict(
      model='pix2pixHD', gpu_ids=[], is_train=True,
      no_label=False, no_label_encoding=True, no_instance=False, no_feat=False,
      no_feat_encoding=True, sem_masking=False, num_labels=35, contain_dontcare_label=False,
      input_nc=3, num_out_channels=3, ngf=64, netG='global', n_downsample_global=4,
      n_blocks_global=9, n_local_enhancers=1, n_blocks_local=3, norm='instance',
      no_generator_binarization=True, bin_generator_before_res=False,
      generator_binarizer_out_channels=128, no_encoder_binarization=True,
      no_label_encoder_binarization=True, no_lsgan=False, ndf=64, n_layers_D=3, num_D=2,
      load_model=False, checkpoints_dir=None, save_dir='./checkpoints', pool_size=0,
      distortion_loss_fn='l1', niter_fix_global=0, lr=2e-4, beta1=0.5, beta2=0.999,
      use_compressed=False, ext='bpg', quality=[42], normalize_mean=[0.5, 0.5, 0.5],
      normalize_std=[1.0, 1.0, 1.0], data_type=32, match_raw_feat=False, zero_vis=False,
      zero_sem=False, zero_ins=False, use_netE_output=False, lambda_feat=10.0,
      lambda_distortion=10.0, anneal_lambda=False, anneal_interval=5000, anneal_factor=5.0,
      no_d_gan_loss=False, no_g_gan_loss=False, no_vgg_loss=False, no_gan_feat_loss=False,
      no_distortion_loss=False, fp16=False, tf_log=False, schedule_lr=False,
      lr_decay_factor=0.1, lr_decay_patience=5, verbose=False, batch_size=1,
      skip_unused_losses=False, vgg19_state_dict=None, vgg_random_init=True, checkpoint_resblocks=False)
  o.update(over)
  return types.SimpleNamespace(**o)


def gen_cfg(opt):
  return dict(netG=opt.netG, ngf=opt.ngf, n_downsample_global=opt.n_downsample_global,
              n_blocks_global=opt.n_blocks_global, n_local_enhancers=opt.n_local_enhancers,
              n_blocks_local=opt.n_blocks_local)


def semantics_nc(opt):
  """pix2pixHD_model.py:118-133 for the no_label_encoding branch."""
  nc = opt.num_labels + 1 if opt.contain_dontcare_label else opt.num_labels
  if not opt.no_instance:
    nc += 1
  return nc


def edge_map(inst):
  """4-neighbour instance-boundary indicator (pix2pixHD_model.py:774-783)."""
  e = torch.zeros(inst.shape, dtype=torch.bool)
  dx = inst[:, :, :, 1:] != inst[:, :, :, :-1]
  dy = inst[:, :, 1:, :] != inst[:, :, :-1, :]
  e[:, :, :, 1:] |= dx
  e[:, :, :, :-1] |= dx
  e[:, :, 1:, :] |= dy
  e[:, :, :-1, :] |= dy
  return e.float()


def preprocess(x_dict, opt):
  """label -> one-hot, instance -> edge, concat (pix2pixHD_model.py:375-396)."""
  lab = x_dict['label'].long()
  b, _, h, w = lab.shape
  nc = opt.num_labels + 1 if opt.contain_dontcare_label else opt.num_labels
  onehot = torch.zeros(b, nc, h, w).scatter_(1, lab, 1.0)
  out = torch.cat((onehot, edge_map(x_dict['instance'])), dim=1)
  return out.to(x_dict['image'].dtype) if 'image' in x_dict else out


def tensor2im(img, opt):
  """De-normalise, clip, truncate to uint8; [B,3,H,W] -> uint8 [B,H,W,3] (misc.py:64-95)."""
  a = img.detach().cpu().float().numpy()
  std = np.asarray(opt.normalize_std)[None, :, None, None]
  mean = np.asarray(opt.normalize_mean)[None, :, None, None]
  a = np.clip((a * std + mean) * 255.0, 0, 255)
  return np.transpose(a, (0, 2, 3, 1)).astype(np.uint8)


class OracleTrainer(object):
  """Same observable behaviour as Pix2PixHDTrainer(opt, 'train') on torch-CPU."""

  def __init__(self, opt, sd_G=None, sd_D=None, sd_vgg=None):
    self.opt = opt
    nc = semantics_nc(opt)
    self.cfg = gen_cfg(opt)
    if sd_G is None:   # same construction order as Pix2PixHDModel.__init__: G, then D
      sd_G = nets.init_generator(self.cfg, nc + opt.input_nc, opt.num_out_channels)
    if sd_D is None:
      sd_D = nets.init_discriminator(nc + opt.num_out_channels, opt.ndf, opt.n_layers_D, opt.num_D)
    if sd_vgg is None:
      sd_vgg = nets.init_vgg19()
    self.G = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd_G.items())
    self.D = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in sd_D.items())
    self.vgg = OrderedDict((k, v.clone()) for k, v in sd_vgg.items())
    g_params = list(self.G.values())
    if opt.niter_fix_global > 0:     # pix2pixHD_model.py:251-268
      g_params = [v for k, v in self.G.items() if k.startswith('model%d' % opt.n_local_enhancers)]
    self.optimizer_G = torch.optim.Adam(g_params, lr=opt.lr, betas=(opt.beta1, opt.beta2))
    self.optimizer_D = torch.optim.Adam(list(self.D.values()), lr=opt.lr,
                                        betas=(opt.beta1, opt.beta2))
    self.lambda_distortion_weight = 1.0
    self.steps_taken = 0
    self.last_losses = None

  # -- forward pieces ------------------------------------------------------
  def _inputs(self, x_dict):
    input_label = preprocess(x_dict, self.opt)
    real = x_dict['image']
    src = x_dict['compressed_img'] if self.opt.use_compressed else real
    return input_label, real, src

  def generate(self, input_label, src):
    return nets.generator(self.G, nets.q(torch.cat((input_label, src), dim=1)), self.cfg)

  def netD(self, x):
    return nets.multiscale_d(self.D, x, self.opt.num_D, self.opt.n_layers_D)

  def train_losses(self, x_dict):
    """The 6-tuple of get_train_loss, in LOSS_NAMES order.

    Reference behaviour kept: with use_compressed only the generator INPUT is the
    decoded frame (_get_img rebinds its local `real_image`, pix2pixHD_model.py:517-518);
    every loss still compares against x_dict['real_image'] (:711, :722, :756, :767).
    """
    opt = self.opt
    input_label, real, src = self._inputs(x_dict)
    fake = self.generate(input_label, src)
    real = nets.q(real)            # no-op unless bf16-storage emulation is on (nets.storage_bf16)
    pred_fake_pool = self.netD(torch.cat((input_label.detach(), fake.detach()), dim=1))
    loss_D_fake = nets.gan_loss(pred_fake_pool, False)
    pred_real = self.netD(torch.cat((input_label.detach(), real.detach()), dim=1))
    loss_D_real = nets.gan_loss(pred_real, True)
    pred_fake = self.netD(torch.cat((input_label, fake), dim=1))
    loss_G_GAN = nets.gan_loss(pred_fake, True)
    loss_feat = 0.0
    for i in range(opt.num_D):
      for j in range(len(pred_fake[i]) - 1):
        loss_feat = loss_feat + (1.0 / opt.num_D) * F.l1_loss(pred_fake[i][j],
                                                              pred_real[i][j].detach())
    loss_vgg = nets.vgg_loss(self.vgg, fake, real)
    if opt.distortion_loss_fn == 'l1':
      loss_dist = F.l1_loss(fake, real)
    else:
      loss_dist = F.mse_loss(fake, real)
    self.last_fake = fake
    return loss_G_GAN, loss_feat, loss_vgg, loss_dist, loss_D_real, loss_D_fake

  def grads_in_dtype(self, x_dict, dtype):
    """(d loss_G / d G-params, d loss_D / d D-params) at the CURRENT weights, evaluated in
    `dtype` (float64 gives the conditioning yardstick the GPU parity tests use: the L1 terms
    have sign() gradients, so two correct fp32 implementations differ by ~sqrt(#sign flips))."""
    opt = self.opt
    keepG, keepD, keepV = self.G, self.D, self.vgg
    try:
      self.G = OrderedDict((k, v.detach().to(dtype).requires_grad_(True)) for k, v in keepG.items())
      self.D = OrderedDict((k, v.detach().to(dtype).requires_grad_(True)) for k, v in keepD.items())
      self.vgg = OrderedDict((k, v.to(dtype)) for k, v in keepV.items())
      xd = {k: (v.to(dtype) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in x_dict.items()}
      L = dict(zip(LOSS_NAMES, self.train_losses(xd)))
      loss_G = 0
      if not opt.no_g_gan_loss: loss_G = loss_G + L['G_GAN']
      if not opt.no_vgg_loss: loss_G = loss_G + L['G_VGG'] * opt.lambda_feat
      if not opt.no_gan_feat_loss: loss_G = loss_G + L['G_GAN_Feat'] * opt.lambda_feat
      if not opt.no_distortion_loss:
        loss_G = loss_G + L['G_Distortion'] * opt.lambda_distortion * self.lambda_distortion_weight
      gG = torch.autograd.grad(loss_G, list(self.G.values()), retain_graph=True, allow_unused=True)
      gD = torch.autograd.grad((L['D_fake'] + L['D_real']) * 0.5, list(self.D.values()), allow_unused=True)
      return (OrderedDict(zip(self.G.keys(), gG)), OrderedDict(zip(self.D.keys(), gD)))
    finally:
      self.G, self.D, self.vgg = keepG, keepD, keepV

  # -- public API mirrored from Pix2PixHDTrainer -----------------------------
  def step(self, x_dict, keep_grads=False):
    opt = self.opt
    L = dict(zip(LOSS_NAMES, self.train_losses(x_dict)))
    zero = lambda: torch.zeros(1, requires_grad=True)
    loss_D = (L['D_fake'] + L['D_real']) * 0.5 if not opt.no_d_gan_loss else zero()
    g_feat = L['G_GAN_Feat'] * opt.lambda_feat if not opt.no_gan_feat_loss else zero()
    g_vgg = L['G_VGG'] * opt.lambda_feat if not opt.no_vgg_loss else zero()
    g_dist = (L['G_Distortion'] * opt.lambda_distortion * self.lambda_distortion_weight
              if not opt.no_distortion_loss else zero())
    g_gan = L['G_GAN'] if not opt.no_g_gan_loss else zero()
    loss_G = g_gan + g_vgg + g_feat + g_dist
    self.last_losses = OrderedDict((k, float(v.detach())) for k, v in L.items())

    self.optimizer_G.zero_grad()
    loss_G.backward()
    if keep_grads:
      self.grads_G = OrderedDict((k, None if v.grad is None else v.grad.clone())
                                 for k, v in self.G.items())
    self.optimizer_G.step()
    self.optimizer_D.zero_grad()   # discards D grads accumulated by loss_G.backward()
    loss_D.backward()
    if keep_grads:
      self.grads_D = OrderedDict((k, None if v.grad is None else v.grad.clone())
                                 for k, v in self.D.items())
    self.optimizer_D.step()
    self.steps_taken += 1
    if opt.anneal_lambda and not (self.steps_taken % opt.anneal_interval):
      self.lambda_distortion_weight *= opt.anneal_factor
    return self.last_losses['G_Distortion']

  def get_img(self, x_dict):
    with torch.no_grad():
      input_label, _, src = self._inputs(x_dict)
      return self.generate(input_label, src)

  def get_eval_loss(self, x_dict):
    """Distortion on de-normalised uint8-truncated images (0..255 scale)."""
    with torch.no_grad():
      recon = self.get_img(x_dict)
      a = torch.tensor(tensor2im(recon, self.opt).transpose(0, 3, 1, 2)).float()
      b = torch.tensor(tensor2im(x_dict['image'], self.opt).transpose(0, 3, 1, 2)).float()
      fn = F.l1_loss if self.opt.distortion_loss_fn == 'l1' else F.mse_loss
      return float(fn(a, b))


def synthetic_batch(batch, height, width, seed=1234, num_labels=35):
  """Cityscapes-shaped synthetic x_dict (SURVEY.md §8d "Synthetic inputs")."""
  g = torch.Generator().manual_seed(seed)
  ch, cw = max(height // 32, 1), max(width // 32, 1)
  lab = torch.randint(0, num_labels, (batch, 1, ch, cw), generator=g)
  inst = (torch.randint(0, 64, (batch, 1, ch, cw), generator=g) * 1000
          + torch.randint(0, 10, (batch, 1, ch, cw), generator=g))
  up = lambda t: F.interpolate(t.float(), size=(height, width), mode='nearest')
  image = torch.rand(batch, 3, height, width, generator=g) - 0.5
  comp = (image + 0.05 * torch.randn(batch, 3, height, width, generator=g)).clamp_(-0.5, 0.5)
  return {'label': up(lab), 'instance': up(inst).long(), 'image': image,
          'compressed_img': comp, 'path': ['synthetic_%d' % i for i in range(batch)]}
