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
      self._dp = None
      if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        self.enable_data_parallel(reduce_dtype=torch.bfloat16 if getattr(opt, 'bf16_grad_reduce', False) else None)

  # ---- data parallelism (no reference counterpart: base_parser.py:234-237 refuses >1 GPU) ----
  def enable_data_parallel(self, bucket_bytes=None, process_group=None, reduce_dtype=None, overlap=None):
    """Replicate-and-average: broadcast rank 0's weights once, re-home every gradient into flat all-reduce buckets, and
    let Adam divide by the world size.  Runs for any initialised process group, world size 1 included (the RCCL
    calls are then no-ops in value, which is how the path is exercised on a one-GPU box).  reduce_dtype
    torch.bfloat16 halves the bytes on the wire (SURVEY.md 8d config 4 reports both)."""
    world = dist.get_world_size(process_group)
    for net in (self.model.netG, self.model.netD):
      for p in net.parameters():
        # 4-D masters are channels_last: hand the collective the dense KRSC view of the same memory
        dense = p.data.permute(0, 2, 3, 1) if p.dim() == 4 else p.data
        assert dense.is_contiguous()
        dist.broadcast(dense, src=0, group=process_group)
    bump_weights_epoch()
    # overlap: where the generator's gradient all-reduce runs.  'd_backward' (default): all of G's buckets start when G's
    # backward has finished and overlap the discriminator's backward; 'layers': each bucket starts from the backward hook
    # of its last layer and overlaps the rest of G's backward (round 1-2 behaviour).  Same results either way.
    overlap = overlap or getattr(self.opt, 'ddp_overlap', 'd_backward')
    assert overlap in ('d_backward', 'layers'), overlap
    if bucket_bytes is None:
      # 'layers': 64 MB buckets (one ResnetBlock filter = 37.7 MB each) so that a bucket goes on the wire as soon as its layer is
      # done; 'd_backward': everything starts at once, so fewer and larger collectives (256 MB: the generator's 730 MB in 3 + the
      # small layers) -- a ring all-reduce's fixed cost (2 (N-1) hops of latency + a kernel launch) is paid per bucket
      bucket_bytes = (64 << 20) if overlap == 'layers' else (256 << 20)
    self._dp = dict(bucket_bytes=bucket_bytes, process_group=process_group, reduce_dtype=reduce_dtype, world=world,
                    overlap=overlap)
    self._rebuild_buckets('G', self.model.netG, self.optimizer_G)
    self._rebuild_buckets('D', self.model.netD, self.optimizer_D)

  def _rebuild_buckets(self, tag, net, optim):
    """(Re)create the gradient buckets of one network over the parameters its optimizer trains, fold 1/world into the
    optimizer and point every conv's gradient-ready hook at the new buckets."""
    dp = self._dp
    trained = {id(p) for grp in optim.param_groups for p in grp['params']}
    named = [(n, p) for n, p in net.named_parameters() if id(p) in trained]
    buckets = GradBuckets(named, bucket_bytes=dp['bucket_bytes'], process_group=dp['process_group'],
                          reduce_dtype=dp['reduce_dtype'], always_reduce=True,
                          defer=(tag == 'G' and dp.get('overlap', 'd_backward') == 'd_backward'))
    self.model.grad_buckets[tag] = buckets
    optim.grad_scale = 1.0 / dp['world']
    for m in net.modules():
      if isinstance(m, HipConv2d):
        m.grad_ready_hook = (lambda mod, b=buckets: (b.mark_ready(mod.weight), b.mark_ready(mod.bias)))

  def update_fixed_params(self):
    """End of the `niter_fix_global` phase (train.py calls model.update_fixed_params and swaps the optimizer,
    pix2pixHD_model.py:795-804): from now on the coarse generator trains too.  The trainer owns the transition so
    that, under data parallelism, the new parameters get gradient buckets, hooks and the 1/world scale as well."""
    self.optimizer_G = self.model.update_fixed_params(self.optimizer_G)
    if getattr(self, '_dp', None) is not None:
      self._rebuild_buckets('G', self.model.netG, self.optimizer_G)
    if getattr(self.opt, 'schedule_lr', False):
      self.scheduler_G = ReduceLROnPlateau(self.optimizer_G, 'min', factor=self.opt.lr_decay_factor,
                                           patience=self.opt.lr_decay_patience)
    return self.optimizer_G

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

  # ---- checkpoints ------------------------------------------------------------------------------
  # File contract of the reference (pix2pixHD_trainer.py:119-176, base_model.py:54-59): <save_dir>/net_G.pth,
  # net_D.pth (state dicts) and stats_and_optim.pt with the keys below.  Optional entries are written / read only when
  # their feature is on, and a checkpoint without them still loads.
  _OPTIONAL_STATE = (('scheduler_G_state_dict', 'schedule_lr'), ('scheduler_D_state_dict', 'schedule_lr'),
                     ('lambda_distortion_weight', 'anneal_lambda'))

  def _is_writer(self):
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0

  def save(self, epoch, val_loss_value):
    """Rank 0 writes (replicas are identical); the other ranks wait at a barrier so that nobody races ahead and
    reads a half-written checkpoint."""
    self.best_val_loss = val_loss_value
    if self._is_writer():
      os.makedirs(self.opt.save_dir, exist_ok=True)
      print('\nwriting checkpoint (epoch %d) to %s\n' % (epoch, self.opt.save_dir))
      record = dict(epoch=epoch, steps_taken=self.steps_taken, best_val_loss=self.best_val_loss,
                    optimizer_G_state_dict=self.optimizer_G.state_dict(),
                    optimizer_D_state_dict=self.optimizer_D.state_dict())
      for key, flag in self._OPTIONAL_STATE:
        if getattr(self.opt, flag, False):
          src = getattr(self, key[:-len('_state_dict')], None) if key.endswith('_state_dict') else getattr(self, key)
          record[key] = src.state_dict() if key.endswith('_state_dict') else src
      torch.save(record, os.path.join(self.opt.save_dir, 'stats_and_optim.pt'))
      self.model.save()
    if dist.is_available() and dist.is_initialized():
      dist.barrier()

  def load(self):
    path = os.path.join(self.opt.checkpoints_dir, 'stats_and_optim.pt')
    where = 'cuda:' + str(self.opt.gpu_ids[0]) if self.model.use_gpu() else 'cpu'
    record = torch.load(path, map_location=where)
    if self.mode != 'train':
      return
    self.optimizer_G.load_state_dict(record['optimizer_G_state_dict'])
    self.optimizer_D.load_state_dict(record['optimizer_D_state_dict'])
    for key, flag in self._OPTIONAL_STATE:
      if not getattr(self.opt, flag, False):
        continue
      if key not in record:
        print('checkpoint has no %s: keeping the fresh one' % key)
      elif key.endswith('_state_dict'):
        getattr(self, key[:-len('_state_dict')]).load_state_dict(record[key])
      else:
        setattr(self, key, record[key])
    self.best_val_loss = record['best_val_loss']
    self.steps_taken = record['steps_taken']
    self.start_epoch = record['epoch'] + 1
    print('\nresumed %s: best val loss %.4f, continuing with epoch %d\n'
          % (path, self.best_val_loss, self.start_epoch + 1))
