"""Trainer contract (reference: ctu/trainers/base_trainer.py:5-87)."""
import os

import torch


class BaseTrainer(torch.nn.Module):

  def __init__(self, opt, mode='train'):
    super(BaseTrainer, self).__init__()
    if mode not in ('train', 'test'):
      raise ValueError('Invalid trainer mode: {}'.format(mode))
    self.opt = opt
    self.mode = mode
    if mode == 'train':
      self.steps_taken = 0       # total number of train steps taken
      self.start_epoch = 0
      self.best_val_loss = 1e12
      self.writer = None
      if getattr(opt, 'tf_log', False):
        # the reference logs through TF1's FileWriter; torch's SummaryWriter writes the same event files
        from torch.utils.tensorboard import SummaryWriter
        self.writer = SummaryWriter(os.path.join(opt.save_dir, 'tf_log'))

  def train(self, mode=True):
    """nn.Module.train walks every submodule -- ~0.35 ms of host time for G + D + VGG19, and `step` calls it first
    thing while the GPU sits idle behind the previous step's loss readback.  Skip the walk when the trainer, its model
    and the model's networks already are in the requested mode (results are unchanged: the walk would set flags that
    are already set)."""
    top = [self] + list(self.children())
    for m in list(top[1:]):
      top.extend(m.children())
    if getattr(self, '_mode_walked', None) == mode and all(m.training == mode for m in top):
      return self
    self._mode_walked = mode
    return super(BaseTrainer, self).train(mode)

  def load(self):
    pass

  def save(self, epoch, val_loss_value):
    pass

  def get_img(self, x_dict):
    raise NotImplementedError

  def step(self, x_dict):
    raise NotImplementedError

  def get_eval_loss(self, x_dict):
    raise NotImplementedError

  def scheduler_step(self, val_loss_value):
    pass

  def log_loss_values(self, loss_dict):
    if self.writer is None:
      return
    for k, v in loss_dict.items():
      self.writer.add_scalar(k, float(v), self.steps_taken)
