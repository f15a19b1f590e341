"""Script-default options and Cityscapes-shaped synthetic inputs for benchmarks and examples.

The reference builds `opt` with its argparse parsers (ctu/parsers/base_parser.py, train_parser.py) and reads batches
from disk; both sit outside the hot path.  This module supplies the same `opt` fields with the values
of scripts/pix2pixHD_bpg_train.sh:5 and a seeded synthetic `x_dict` of the schema of ctu/data/ctu_dataset.py:93-128
(SURVEY.md §8d "Synthetic inputs").  Product code: it does not import the test oracle.
"""
import types

import torch
import torch.nn.functional as F


def default_opt(**over):
  """The `opt` fields Pix2PixHDModel / Pix2PixHDTrainer read, script defaults."""
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


def synthetic_batch(batch, height, width, seed=1234, num_labels=35):
  """Piecewise-constant label / instance maps on a coarse grid (nearest-upsampled: realistic region structure),
  uniform image in [-0.5, 0.5), and a noisy copy standing in for the decoded base-codec frame."""
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
