"""CPU-only checks of the host logic added in round 2 (no GPU, no kernel launches): the option setter against the
flag table dumped from the reference, the batched / prefetched codec hand-off against the reference's per-image file
round trip, FusedAdam's checkpoint re-layout, bench.py's argument handling for N > 1, gradient buckets on a bf16 wire."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- option setter (pix2pixHD_model.py:21-102) ---------------------------------------------------------------------
def test_option_setter_is_superset_of_reference(golden_dir):
  """tests/golden/option_setter_flags.json = the reference setter's flags (oracle/make_option_fixture.py).  Every one of
  them must exist here with the same default, type, choices and action kind, so that a reference command line or opt.pkl
  parses unchanged."""
  from ctu.models import get_option_setter
  with open(os.path.join(golden_dir, 'option_setter_flags.json')) as fh:
    ref = json.load(fh)['flags']
  assert len(ref) >= 59
  parser = argparse.ArgumentParser(add_help=False)
  get_option_setter('pix2pixHD')(parser, True)
  mine = {a.dest: a for a in parser._actions}
  for row in ref:
    assert row['dest'] in mine, 'reference flag %s is missing from the product setter' % row['flags']
    a = mine[row['dest']]
    assert list(a.option_strings) == row['flags']
    assert a.default == row['default'], (row['dest'], a.default, row['default'])
    assert (a.type.__name__ if a.type is not None else None) == row['type'], row['dest']
    assert (list(a.choices) if a.choices is not None else None) == row['choices'], row['dest']
    assert type(a).__name__ == row['action'], row['dest']
  # the full script command line of scripts/pix2pixHD_bpg_train.sh:5 (model flags) parses
  ns = parser.parse_args('--no_label_encoding --no_feat_encoding --no_generator_binarization --use_compressed --ext bpg '
                         '--quality 42 --netG local --ngf 32 --niter_fix_global 20 --binary_mask --use_dropout'.split())
  assert ns.netG == 'local' and ns.ngf == 32 and ns.use_dropout and ns.vgg19_state_dict is None


# ---- codec hand-off (pix2pixHD_model.py:287-359) -------------------------------------------------------------------
def _reference_style_round_trip(image, opt, tmp_dir):
  """The reference's own sequence for ONE image (pix2pixHD_model.py:336-351): tensor2im -> PNG file -> converter ->
  Image.open -> ToTensor -> Normalize, restated with numpy/PIL (torchvision is not installed here)."""
  from PIL import Image
  from ctu.utils import codec
  from ctu.utils.misc import tensor2im
  img = Image.fromarray(tensor2im(image, opt).squeeze())
  name = os.path.join(tmp_dir, 'tmp_image.png')
  img.save(name)
  out = codec.converter(name, opt.ext, opt.quality[0])
  arr = np.asarray(Image.open(out).convert('RGB'))
  t = torch.from_numpy(arr).permute(2, 0, 1).float().div(255)                       # ToTensor
  mean = torch.tensor(opt.normalize_mean)[:, None, None]
  std = torch.tensor(opt.normalize_std)[:, None, None]
  return ((t - mean) / std).unsqueeze(0)                                             # Normalize


@pytest.mark.parametrize('ext,quality', [('jpg', 42), ('webp', 60)])
def test_codec_collate_batch4_equals_per_image_loop(tmp_path, ext, quality):
  from ctu.utils import codec
  from ctu.utils.synthetic import default_opt, synthetic_batch
  opt = default_opt(use_compressed=True, ext=ext, quality=[quality])
  xd = synthetic_batch(4, 64, 96, seed=5)
  samples = [{k: (v[i] if torch.is_tensor(v) else v[i]) for k, v in xd.items() if k != 'compressed_img'} for i in range(4)]
  batch = codec.CodecCollate(opt)(samples)
  assert batch['compressed_img'].shape == (4, 3, 64, 96) and batch['image'].shape == (4, 3, 64, 96)
  assert torch.equal(batch['image'], xd['image']) and torch.equal(batch['label'], xd['label'])
  for i in range(4):
    want = _reference_style_round_trip(xd['image'][i:i + 1], opt, str(tmp_path))
    assert torch.equal(batch['compressed_img'][i:i + 1], want), 'image %d differs from the per-image file round trip' % i
  # lossy: the decoded frame differs from the original, but not wildly
  err = (batch['compressed_img'] - xd['image']).abs().mean().item()
  assert 1e-4 < err < 0.25


def test_codec_collate_runs_in_dataloader_workers():
  """The collate (and with it the codec) executes in the worker processes, ahead of consumption."""
  from torch.utils.data import DataLoader, Dataset
  from ctu.utils import codec
  from ctu.utils.synthetic import default_opt, synthetic_batch
  opt = default_opt(use_compressed=True, ext='jpg', quality=[50])

  class DS(Dataset):
    def __len__(self):
      return 8

    def __getitem__(self, i):
      xd = synthetic_batch(1, 32, 48, seed=100 + i)
      return {'label': xd['label'][0], 'instance': xd['instance'][0], 'image': xd['image'][0], 'path': 'img%d' % i,
              'worker_pid': os.getpid()}

  loader = DataLoader(DS(), batch_size=4, num_workers=2, prefetch_factor=2, collate_fn=codec.CodecCollate(opt))
  seen = 0
  for batch in loader:
    assert batch['compressed_img'].shape == (4, 3, 32, 48)
    assert int(batch['worker_pid'][0]) != os.getpid()
    ref = codec.compress_images(batch['image'], opt)
    assert torch.equal(batch['compressed_img'], ref)
    seen += 1
  assert seen == 2


def test_private_tmp_dirs_do_not_collide(tmp_path):
  from ctu.utils import codec
  d1 = codec.private_tmp_dir(str(tmp_path))
  assert os.path.isdir(d1) and str(os.getpid()) in os.path.basename(d1)
  assert codec.private_tmp_dir(str(tmp_path)) == d1          # stable within a process
  code = ('import sys; sys.path.insert(0, %r); from ctu.utils import codec; print(codec.private_tmp_dir(%r))'
          % (os.path.join(ROOT, 'jpd-se_amd'), str(tmp_path)))
  other = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, text=True, check=True).stdout.strip()
  assert other != d1 and os.path.dirname(other) == os.path.dirname(d1)


# ---- FusedAdam checkpoint interop (pix2pixHD_trainer.py:124-136,147-148) -------------------------------------------
def test_fused_adam_loads_torch_adam_state_from_contiguous_parameters():
  from jpdse_hip.optim import FusedAdam
  w = torch.nn.Parameter(torch.randn(6, 4, 3, 3).contiguous(memory_format=torch.channels_last))
  b = torch.nn.Parameter(torch.randn(6))
  rw, rb = torch.nn.Parameter(w.detach().clone().contiguous()), torch.nn.Parameter(b.detach().clone())
  ref = torch.optim.Adam([rw, rb], lr=2e-4, betas=(0.5, 0.999))
  rw.grad, rb.grad = torch.randn_like(rw), torch.randn_like(rb)
  ref.step()
  ref.step()
  import copy
  sd = copy.deepcopy(ref.state_dict())
  sd['state'][1]['step'] = 2                                  # checkpoints of old torch versions keep an int
  opt = FusedAdam([w, b], lr=1.0)
  opt.load_state_dict(sd)
  assert opt.param_groups[0]['lr'] == 2e-4 and tuple(opt.param_groups[0]['betas']) == (0.5, 0.999)
  for p, r in ((w, rw), (b, rb)):
    st = opt.state[p]
    assert st['exp_avg'].stride() == p.stride() and st['exp_avg_sq'].stride() == p.stride()
    assert torch.equal(st['exp_avg'], ref.state[r]['exp_avg']) and torch.equal(st['exp_avg_sq'], ref.state[r]['exp_avg_sq'])
    assert torch.is_tensor(st['step']) and float(st['step']) == 2.0
  out = opt.state_dict()                                       # and it round-trips into torch.optim.Adam again
  back = torch.optim.Adam([torch.nn.Parameter(rw.detach().clone()), torch.nn.Parameter(rb.detach().clone())])
  back.load_state_dict(out)


# ---- bench.py launcher contract -------------------------------------------------------------------------------------
def test_bench_gpus_n_without_launcher_fails_cleanly_when_gpus_are_missing():
  """`python bench.py --gpus 2` (the driver's direct form) must not die on an assertion: it self-launches N ranks, and
  on a box with fewer GPUs it says so and exits 2 before touching any device."""
  env = dict(os.environ)
  env.pop('WORLD_SIZE', None)
  # hide every GPU from the child: "fewer GPUs than ranks" must hold on any box (on a multi-GPU node the unhidden child
  # would really launch a 2-rank benchmark from the CPU suite)
  env['HIP_VISIBLE_DEVICES'] = env['ROCR_VISIBLE_DEVICES'] = env['CUDA_VISIBLE_DEVICES'] = ''
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
  assert r.returncode == 2, r.stderr[-2000:]
  assert 'AssertionError' not in r.stderr and 'GPU(s) visible' in r.stderr
  env['WORLD_SIZE'], env['RANK'], env['LOCAL_RANK'] = '4', '0', '0'
  r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], stdout=subprocess.PIPE,
                     stderr=subprocess.PIPE, text=True, env=env, timeout=300)
  assert r.returncode == 2 and 'WORLD_SIZE=4 but --gpus 2' in r.stderr
