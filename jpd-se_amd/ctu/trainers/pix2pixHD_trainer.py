"""JPD-SE trainer on the MI355X-native model (reference: ctu/trainers/pix2pixHD_trainer.py).

`step()` keeps the reference's observable contract -- same loss weighting (:48-56), same
update order G then D (:64-78), returns the G_Distortion float (:85) -- but runs the explicit
HIP schedule of `Pix2PixHDModel.train_step` with a single host sync, and adds what the
reference lacks: per-image data parallelism over torch.distributed (RCCL) when a process
group is initialised."""
import os

import torch
import torch.distributed as dist
from torch.optim.lr_scheduler import ReduceLROnPlateau

from ctu.models.pix2pixHD_model import Pix2PixHDModel
from ctu.trainers.base_trainer import BaseTrainer
from jpdse_hip.ddp import GradBuckets
from jpdse_hip.layers import HipConv2d, bump_weights_epoch


class Pix2PixHDTrainer(BaseTrainer):

  def __init__(self, opt, mode='train'):
    super(Pix2PixHDTrainer, self).__init__(opt, mode)
    self.model = Pix2PixHDModel(opt)
    self.print_losses = getattr(opt, 'print_losses', True)
    if mode == 'train':
      self.optimizer_G, self.optimizer_D = self.model.create_optimizers(opt)
      if getattr(opt, 'schedule_lr', False):
        self.scheduler_G = ReduceLROnPlateau(self.optimizer_G, 'min', factor=opt.lr_decay_factor,
                                             patience=opt.lr_decay_patience)
        self.scheduler_D = ReduceLROnPlateau(self.optimizer_D, 'min', factor=opt.lr_decay_factor,
                                             patience=opt.lr_decay_patience)
      self.lambda_distortion_weight = 1.
      if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        self.enable_data_parallel()

  # ---- data parallelism (no reference counterpart: base_parser.py:234-237 refuses >1 GPU) ----
  def enable_data_parallel(self, bucket_bytes=64 << 20, process_group=None):
    """Replicate-and-average: broadcast rank 0's weights once, re-home every gradient into flat
    all-reduce buckets, and let Adam divide by the world size."""
    world = dist.get_world_size(process_group)
    for net in (self.model.netG, self.model.netD):
      for p in net.parameters():
        # 4-D masters are channels_last: hand the collective the dense KRSC view of the same memory
        dense = p.data.permute(0, 2, 3, 1) if p.dim() == 4 else p.data
        assert dense.is_contiguous()
        dist.broadcast(dense, src=0, group=process_group)
    bump_weights_epoch()
    for tag, net, optim in (('G', self.model.netG, self.optimizer_G), ('D', self.model.netD, self.optimizer_D)):
      trained = {id(p) for grp in optim.param_groups for p in grp['params']}
      named = [(n, p) for n, p in net.named_parameters() if id(p) in trained]
      buckets = GradBuckets(named, bucket_bytes=bucket_bytes, process_group=process_group)
      self.model.grad_buckets[tag] = buckets
      optim.grad_scale = 1.0 / world
      for m in net.modules():
        if isinstance(m, HipConv2d):
          m.grad_ready_hook = (lambda mod, b=buckets: (b.mark_ready(mod.weight), b.mark_ready(mod.bias)))

  def scheduler_step(self, val_loss_value):
    self.scheduler_G.step(val_loss_value)
    self.scheduler_D.step(val_loss_value)

  def step(self, x_dict):
    self.train()
    L = self.model.train_step(x_dict, self.optimizer_G, self.optimizer_D, self.lambda_distortion_weight)
    self.last_losses = L
    if self.print_losses:
      print('g_gan: {:.4f}, g_gan_feat_match: {:.4f}, g_vgg: {:.4f}, g_distortion ({}): {:.4f}, d_real: {:.4f}, '
            'd_fake: {:.4f}'.format(L['G_GAN'], L['G_GAN_Feat'], L['G_VGG'], self.opt.distortion_loss_fn,
                                    L['G_Distortion'], L['D_real'], L['D_fake']))
    self.steps_taken += 1
    if self.opt.anneal_lambda and not (self.steps_taken % self.opt.anneal_interval):
      self.lambda_distortion_weight *= self.opt.anneal_factor
    if getattr(self.opt, 'tf_log', False):
      self.log_loss_values(L)
    return L['G_Distortion']

  def get_eval_loss(self, x_dict):
    self.eval()
    return self.model(x_dict, self.opt, mode='get_eval_loss').item()

  def get_code(self, x_dict):
    return self.model(x_dict, self.opt, mode='get_code')

  def get_eval_rate(self, x_dict):
    return self.model(x_dict, self.opt, mode='get_eval_rate')

  def get_img(self, x_dict):
    self.eval()
    return self.model(x_dict, self.opt, mode='get_img')

  # ---- checkpoints (stats_and_optim.pt + net_{G,D}.pth: pix2pixHD_trainer.py:119-176) --------
  def save(self, epoch, val_loss_value):
    self.best_val_loss = val_loss_value
    os.makedirs(self.opt.save_dir, exist_ok=True)
    print('\nsaving checkpoints to {}...\n'.format(self.opt.save_dir))
    states = {'epoch': epoch, 'steps_taken': self.steps_taken,
              'optimizer_G_state_dict': self.optimizer_G.state_dict(),
              'optimizer_D_state_dict': self.optimizer_D.state_dict(),
              'best_val_loss': self.best_val_loss}
    if getattr(self.opt, 'schedule_lr', False):
      states['scheduler_G_state_dict'] = self.scheduler_G.state_dict()
      states['scheduler_D_state_dict'] = self.scheduler_D.state_dict()
    if self.opt.anneal_lambda:
      states['lambda_distortion_weight'] = self.lambda_distortion_weight
    torch.save(states, os.path.join(self.opt.save_dir, 'stats_and_optim.pt'))
    self.model.save()
    print('\ncheckpoint saved!\n')

  def load(self):
    print('\nloading checkpoints from {}...\n'.format(self.opt.checkpoints_dir))
    path = os.path.join(self.opt.checkpoints_dir, 'stats_and_optim.pt')
    where = 'cuda:' + str(self.opt.gpu_ids[0]) if self.model.use_gpu() else 'cpu'
    saved = torch.load(path, map_location=where)
    if self.mode == 'train':
      self.optimizer_G.load_state_dict(saved['optimizer_G_state_dict'])
      self.optimizer_D.load_state_dict(saved['optimizer_D_state_dict'])
      if getattr(self.opt, 'schedule_lr', False):
        if 'scheduler_G_state_dict' in saved:
          self.scheduler_G.load_state_dict(saved['scheduler_G_state_dict'])
          self.scheduler_D.load_state_dict(saved['scheduler_D_state_dict'])
        else:
          print('Did not find scheduler state dicts from checkpoint. Not loading them...')
      self.best_val_loss = saved['best_val_loss']
      self.steps_taken = saved['steps_taken']
      if self.opt.anneal_lambda:
        if 'lambda_distortion_weight' in saved:
          self.lambda_distortion_weight = saved['lambda_distortion_weight']
        else:
          print('Did not find lambda distortion weight from checkpoint. Not loading it...')
      self.start_epoch = saved['epoch'] + 1
      print('\ncurrent best val loss: {:.4f}\n'.format(self.best_val_loss))
      print('\nnow starting from epoch {}...\n'.format(self.start_epoch + 1))
    print('\ncheckpoint loaded!\n')
