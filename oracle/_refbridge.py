"""Build-container-only bridge to the real reference (TEST INFRASTRUCTURE).

Imports `/root/reference` in-process exactly as SURVEY.md §8c describes: an inert
in-memory `torchvision` placeholder is registered first because the reference's
networks.py:473 does `from torchvision import models` at module level and
torchvision is not installed.  The placeholder's `models.vgg19()` returns a
locally built VGG19-'E' feature stack carrying the oracle's seeded weights
(`oracle.ctu_cpu.nets.init_vgg19`), since the ImageNet checkpoint is a download.

`/root/reference` does not exist on the GPU box: nothing under tests/ marked
gpu, smoke() or bench.py may import this module.
"""
import os
import sys
import types

import torch

REFERENCE_ROOT = '/root/reference'


def available():
  return os.path.isdir(os.path.join(REFERENCE_ROOT, 'ctu'))


def _vgg_features(sd):
  from oracle.ctu_cpu import nets
  layers, cin, ci = [], 3, 0
  for item in nets.VGG_CFG + ('M',):   # torchvision's cfg 'E' ends with a pool; unused by [0:30]
    if item == 'M':
      layers.append(torch.nn.MaxPool2d(2, 2))
      continue
    conv = torch.nn.Conv2d(cin, item, 3, padding=1)
    with torch.no_grad():
      conv.weight.copy_(sd['vgg.%d.weight' % ci])
      conv.bias.copy_(sd['vgg.%d.bias' % ci])
    layers += [conv, torch.nn.ReLU(inplace=True)]
    cin, ci = item, ci + 1
  return torch.nn.Sequential(*layers)


def import_reference(vgg_sd=None):
  """Returns the reference's (networks module, Pix2PixHDModel, Pix2PixHDTrainer)."""
  if not available():
    raise RuntimeError('reference tree not present (expected only in the build container)')
  from oracle.ctu_cpu import nets
  if vgg_sd is None:
    vgg_sd = nets.init_vgg19()
  if 'torchvision' not in sys.modules:
    tv = types.ModuleType('torchvision')
    tv.models = types.ModuleType('torchvision.models')
    tv.transforms = types.ModuleType('torchvision.transforms')
    sys.modules['torchvision'] = tv
    sys.modules['torchvision.models'] = tv.models
    sys.modules['torchvision.transforms'] = tv.transforms
  tvm = sys.modules['torchvision.models']

  def vgg19(pretrained=False, **_):
    state = torch.get_rng_state()       # building the stand-in must not move the RNG
    holder = types.SimpleNamespace(features=_vgg_features(vgg_sd))
    torch.set_rng_state(state)
    return holder
  tvm.vgg19 = vgg19
  if REFERENCE_ROOT not in sys.path:
    sys.path.insert(0, REFERENCE_ROOT)
  from ctu.models.pix2pixHD_networks import networks
  from ctu.models.pix2pixHD_model import Pix2PixHDModel
  from ctu.trainers.pix2pixHD_trainer import Pix2PixHDTrainer
  return networks, Pix2PixHDModel, Pix2PixHDTrainer
