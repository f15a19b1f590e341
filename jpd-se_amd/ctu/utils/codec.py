"""Host side of the external-codec hand-off: the base codec (BPG / JPEG / WebP / J2K) is third-party
CPU code; what belongs to the hot path is getting its decoded frames to the device as `x_dict['compressed_img']`
[B,3,H,W] without stalling the training thread.

Reference behaviour (ctu/models/pix2pixHD_model.py:287-359): inside `preprocess`, on the training thread, batch size 1
only (`.squeeze()` + `Image.fromarray`), through ONE temp file name per process tree and `subprocess.run(shell=True)`.
Here the same per-image round trip (`round_trip`, same arithmetic: tensor2im -> codec -> ToTensor -> Normalize) is
  * batched: `compress_images` loops over the batch;
  * prefetched: `CodecCollate` is a DataLoader `collate_fn`, so with num_workers > 0 the codec runs in the worker
    PROCESSES, `prefetch_factor` batches ahead of the step that consumes them;
  * collision free: PIL codecs run in memory; BPG (needs files for bpgenc / bpgdec) uses a private mkdtemp directory
    per process (the reference's docstring asks for "a different tmp_folder for each running process").
`Pix2PixHDModel.compress` keeps the synchronous call as the fallback when a batch arrives without 'compressed_img'.
"""
import io
import os
import subprocess
import tempfile

import numpy as np
import torch

from ctu.utils.misc import tensor2im

PIL_FORMATS = {'jpg': 'JPEG', 'webp': 'WEBP', 'j2k': 'JPEG2000'}
_private_dir = {}


def private_tmp_dir(root=None):
  """A directory only this process writes to (created on first use; keyed by pid so that forked DataLoader workers
  do not inherit their parent's)."""
  pid = os.getpid()
  d = _private_dir.get(pid)
  if d is None or not os.path.isdir(d):
    if root is not None:
      os.makedirs(root, exist_ok=True)
    d = tempfile.mkdtemp(prefix='jpdse_codec_%d_' % pid, dir=root)
    _private_dir.clear()
    _private_dir[pid] = d
  return d


def converter(filename, ext, quality):
  """File-based round trip of the reference (pix2pixHD_model.py:287-321): returns the path of the DECODABLE result.
  BPG shells out to bpgenc / bpgdec (no shell, argument list)."""
  from PIL import Image
  stem = os.path.splitext(filename)[0]
  out = stem + '.' + ext
  if ext in ('jpg', 'webp'):
    Image.open(filename).save(out, quality=quality)
    return out
  if ext == 'j2k':
    Image.open(filename).save(out, quality_mode='rates', quality_layers=[quality])
    return out
  if ext == 'bpg':
    decoded = stem + '_decoded_from_bpg.png'
    subprocess.run(['bpgenc', '-q', str(quality), '-o', out, filename], check=True)
    subprocess.run(['bpgdec', '-o', decoded, out], check=True)
    return decoded
  raise ValueError('format must be one of jpg, webp, j2k, or bpg')


def round_trip(img_u8, ext, quality, tmp_dir=None):
  """uint8 HWC image -> the codec's decoded uint8 HWC image.  PIL codecs encode / decode in memory (the same encoder
  call as `converter`, so the same bytes); BPG goes through files in this process's private directory."""
  from PIL import Image
  if ext in PIL_FORMATS:
    buf = io.BytesIO()
    kw = dict(quality_mode='rates', quality_layers=[quality]) if ext == 'j2k' else dict(quality=quality)
    Image.fromarray(img_u8).save(buf, format=PIL_FORMATS[ext], **kw)
    buf.seek(0)
    return np.asarray(Image.open(buf).convert('RGB'))
  if ext == 'bpg':
    d = private_tmp_dir(tmp_dir)
    name = os.path.join(d, 'tmp_image.png')
    Image.fromarray(img_u8).save(name)
    return np.asarray(Image.open(converter(name, ext, quality)).convert('RGB'))
  raise ValueError('format must be one of jpg, webp, j2k, or bpg')


def compress_images(image, opt, tmp_dir=None):
  """image: float [B,3,H,W] normalised like the loader's output -> decoded frames, same shape and normalisation
  (tensor2im -> codec -> ToTensor (/255 in fp32) -> Normalize, as pix2pixHD_model.py:336-351 does for one image)."""
  quality = opt.quality[0] if isinstance(opt.quality, (list, tuple)) else int(opt.quality)
  imgs = tensor2im(image, opt)
  if imgs.ndim == 3:
    imgs = imgs[None]
  mean = torch.tensor(opt.normalize_mean, dtype=torch.float32)[:, None, None]
  std = torch.tensor(opt.normalize_std, dtype=torch.float32)[:, None, None]
  out = []
  for b in range(imgs.shape[0]):
    dec = round_trip(np.ascontiguousarray(imgs[b]), opt.ext, quality, tmp_dir)
    t = torch.from_numpy(dec.astype(np.float32) / np.float32(255.0)).permute(2, 0, 1)
    out.append((t - mean) / std)
  return torch.stack(out, 0)


class CodecCollate(object):
  """DataLoader collate_fn: default collation of the dataset's x_dict samples (ctu/data/ctu_dataset.py:124-128) plus
  'compressed_img' [B,3,H,W], computed where the collate runs -- in the worker processes when num_workers > 0.

      loader = DataLoader(dataset, batch_size=4, num_workers=4, prefetch_factor=2, collate_fn=CodecCollate(opt))
  """

  def __init__(self, opt, tmp_dir=None, base_collate=None):
    self.opt, self.tmp_dir, self.base_collate = opt, tmp_dir, base_collate

  def __call__(self, samples):
    from torch.utils.data import default_collate
    batch = (self.base_collate or default_collate)(samples)
    if getattr(self.opt, 'use_compressed', False) and 'compressed_img' not in batch:
      batch['compressed_img'] = compress_images(batch['image'], self.opt, self.tmp_dir)
    return batch
