"""Checkpoint I/O shared by the models (reference: ctu/models/pix2pixHD_networks/base_model.py).

Files are plain `state_dict`s named net_<label>.pth with the reference's keys and logical
OIHW / IOHW shapes, so checkpoints move freely between the reference and this implementation
(the channels_last memory order used on the device is invisible to torch.save/load)."""
import os

import torch


class BaseModel(torch.nn.Module):

  def __init__(self, opt):
    super(BaseModel, self).__init__()
    self.opt = opt
    self.gpu_ids = opt.gpu_ids
    self.is_train = opt.is_train

  def save_network(self, network, network_label, opt):
    path = os.path.join(opt.save_dir, 'net_%s.pth' % network_label)
    cpu_state = {k: v.detach().cpu().contiguous() for k, v in network.state_dict().items()}
    torch.save(cpu_state, path)

  def load_network(self, network, network_label, opt):
    """Tolerant loader (base_model.py:62-97): exact match, else the subset of matching keys,
    else every tensor whose key and shape match, reporting what stayed uninitialised."""
    path = os.path.join(opt.checkpoints_dir, 'net_%s.pth' % network_label)
    if not os.path.isfile(path):
      print('%s does not exist' % path)
      if network_label == 'G':
        raise FileNotFoundError('generator checkpoint must exist: %s' % path)
      return
    saved = torch.load(path, map_location='cpu')
    own = network.state_dict()
    if set(saved.keys()) == set(own.keys()) and all(saved[k].shape == own[k].shape for k in own):
      network.load_state_dict(saved)
      return
    usable = {k: v for k, v in saved.items() if k in own and v.shape == own[k].shape}
    missing = sorted({k.split('.')[0] for k in own if k not in usable})
    if len(usable) == len(own):
      print('pretrained network %s has excessive layers. Only loading layers that are used' % network_label)
    else:
      print('pretrained network %s has fewer layers. The following are not initialized:' % network_label)
      print(missing)
    merged = dict(own)
    merged.update(usable)
    network.load_state_dict(merged)
