"""Host-side image helpers used by the eval loss and the codec hook."""
import numpy as np
import torch


def tensor2im(image_tensor, opt, imtype=np.uint8, normalize=True, tile=False):
  """[B,C,H,W] or [C,H,W] tensor -> uint8 HWC array(s): de-normalise with opt.normalize_std /
  opt.normalize_mean, scale to 0..255, clip, truncate (reference: ctu/utils/misc.py:64-95)."""
  if isinstance(image_tensor, list):
    return [tensor2im(t, opt, imtype, normalize) for t in image_tensor]
  if tile:
    raise NotImplementedError('tiling is a visualisation feature, outside the hot path')
  t = image_tensor.detach().cpu().float()
  single = t.dim() == 3
  if t.dim() == 2:
    t, single = t[None], True
  if single:
    t = t[None]
  a = t.numpy()
  if normalize:
    std = np.asarray(opt.normalize_std, dtype=np.float64)[None, :, None, None]
    mean = np.asarray(opt.normalize_mean, dtype=np.float64)[None, :, None, None]
    a = a * std + mean
  a = np.clip(np.transpose(a, (0, 2, 3, 1)) * 255.0, 0, 255)
  if a.shape[3] == 1:
    a = a[:, :, :, 0]
  a = a.astype(imtype)
  return a[0] if single else a
