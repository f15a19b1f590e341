"""GPU parity, train-step level: `ctu.trainers.get_trainer(opt)(opt,'train').step()` on the
HIP path against (a) golden step records from the REAL reference and (b) the oracle trainer
stepped side by side; plus get_img / get_eval_loss, checkpoint round trip, bf16 sanity."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import jpdse_hip
from ctu.trainers import get_trainer
from oracle.ctu_cpu import model as omodel
from hip_util import assert_close, rel_err

WEIGHT_TOL = 2e-3     # parameters after Adam steps (fp32); dead biases excluded
LOSS_TOL = 1e-3


def _opts(**kw):
  o = omodel.default_opt(gpu_ids=[0], print_losses=False, **kw)
  return o


def _paired(opt_kw, seed=1234):
  """HIP trainer and oracle trainer holding the same seeded weights."""
  opt = _opts(**opt_kw)
  torch.manual_seed(seed)
  ora = omodel.OracleTrainer(omodel.default_opt(**opt_kw))
  Trainer = get_trainer(opt)
  assert Trainer.__name__ == 'Pix2PixHDTrainer'
  tr = Trainer(opt, 'train')
  tr.model.netG.load_state_dict({k: v.detach() for k, v in ora.G.items()})
  tr.model.netD.load_state_dict({k: v.detach() for k, v in ora.D.items()})
  return tr, ora, opt


def _check_weights(tr, ora, tol, what):
  worst = 0.0
  for k, v in tr.model.netG.state_dict().items():
    if k.endswith('.weight'):
      worst = max(worst, rel_err(v.cpu(), ora.G[k].detach()))
  for k, v in tr.model.netD.state_dict().items():
    if k.endswith('.weight'):
      worst = max(worst, rel_err(v.cpu(), ora.D[k].detach()))
  assert worst <= tol, '%s: weights drifted %.3e' % (what, worst)


def _golden_steps(golden_dir, name):
  g = np.load(os.path.join(golden_dir, name + '.npz'))
  kw = dict(netG=str(g['opt_netG']), ngf=int(g['opt_ngf']), ndf=int(g['opt_ndf']),
            n_blocks_global=int(g['opt_n_blocks_global']))
  tr, ora, opt = _paired(kw, seed=int(g['seed']))
  assert list(tr.model.netG.state_dict().keys()) == list(g['Gkeys'])
  assert list(tr.model.netD.state_dict().keys()) == list(g['Dkeys'])
  b, h, w = int(g['batch']), int(g['height']), int(g['width'])
  for s in range(int(g['steps'])):
    xd = omodel.synthetic_batch(b, h, w, seed=100 + s, num_labels=opt.num_labels)
    ret = tr.step(xd)
    got = [tr.last_losses[k] for k in omodel.LOSS_NAMES]
    np.testing.assert_allclose(got, g['losses:%d' % s], rtol=LOSS_TOL * (1 + s), err_msg='%s step %d' % (name, s))
    np.testing.assert_allclose(ret, float(g['ret:%d' % s]), rtol=LOSS_TOL * (1 + s))
    wmask = np.array([k.endswith('.weight') for k in g['Gkeys']])
    norms = np.array([float(v.double().norm()) for v in tr.model.netG.state_dict().values()])
    np.testing.assert_allclose(norms[wmask], g['Gnorm:%d' % s][wmask], rtol=1e-3)
    wmask = np.array([k.endswith('.weight') for k in g['Dkeys']])
    norms = np.array([float(v.double().norm()) for v in tr.model.netD.state_dict().values()])
    np.testing.assert_allclose(norms[wmask], g['Dnorm:%d' % s][wmask], rtol=1e-3)
    ora.step(xd)
    _check_weights(tr, ora, WEIGHT_TOL * (1 + s), '%s step %d' % (name, s))
  xd = omodel.synthetic_batch(b, h, w, seed=999, num_labels=opt.num_labels)
  img = tr.get_img(xd)
  assert img.shape == (b, 3, h, w) and img.is_cuda
  assert_close(img.cpu(), g['get_img'], 5e-3, name + ' get_img')
  np.testing.assert_allclose(tr.get_eval_loss(xd), float(g['get_eval_loss']), rtol=5e-3)
  return tr


def test_steps_global_generator_golden(golden_dir):
  _golden_steps(golden_dir, 'step_global_ngf8')


def test_steps_local_enhancer_golden(golden_dir):
  _golden_steps(golden_dir, 'step_local_ngf4')


def test_steps_full_width_golden(golden_dir):
  _golden_steps(golden_dir, 'step_global_ngf64_full')


def test_step_vs_oracle_compressed_input_and_mse():
  """use_compressed: the decoded frame feeds G, every loss still compares with the original
  (pix2pixHD_model.py:517-518 vs :711-767); distortion_loss_fn = mse; batch 3, ragged size."""
  kw = dict(ngf=8, ndf=8, n_blocks_global=1, use_compressed=True, distortion_loss_fn='mse')
  tr, ora, opt = _paired(kw)
  for s in range(2):
    xd = omodel.synthetic_batch(3, 48, 80, seed=7 + s)
    tr.step(xd)
    ora.step(xd)
    for k in omodel.LOSS_NAMES:
      assert abs(tr.last_losses[k] - ora.last_losses[k]) <= LOSS_TOL * (1 + s) * max(abs(ora.last_losses[k]), 1e-6), \
          (s, k, tr.last_losses[k], ora.last_losses[k])
    _check_weights(tr, ora, WEIGHT_TOL * (1 + s), 'compressed/mse step %d' % s)


def test_loss_flags_zero_terms():
  """--no_*_loss flags drop terms from the optimised objective (pix2pixHD_trainer.py:48-56)."""
  kw = dict(ngf=8, ndf=8, n_blocks_global=1, no_g_gan_loss=True, no_d_gan_loss=True, no_vgg_loss=True,
            no_gan_feat_loss=True)
  tr, ora, opt = _paired(kw)
  d_before = {k: v.clone() for k, v in tr.model.netD.state_dict().items()}
  xd = omodel.synthetic_batch(1, 32, 64, seed=3)
  tr.step(xd)
  ora.step(xd)
  _check_weights(tr, ora, WEIGHT_TOL, 'phase-3 flags')
  for k, v in tr.model.netD.state_dict().items():
    assert torch.equal(v, d_before[k]), 'D must not move when no_d_gan_loss is set'


def test_checkpoint_roundtrip_reference_layout(tmp_path):
  kw = dict(ngf=8, ndf=8, n_blocks_global=1)
  tr, ora, opt = _paired(kw)
  xd = omodel.synthetic_batch(1, 32, 64, seed=1)
  tr.step(xd)
  tr.opt.save_dir = str(tmp_path)
  tr.save(0, 1.0)
  sd = torch.load(os.path.join(str(tmp_path), 'net_G.pth'))
  assert list(sd.keys()) == list(ora.G.keys())
  assert all(sd[k].shape == ora.G[k].shape and sd[k].is_contiguous() for k in sd)
  so = torch.load(os.path.join(str(tmp_path), 'stats_and_optim.pt'))
  assert {'epoch', 'steps_taken', 'optimizer_G_state_dict', 'optimizer_D_state_dict', 'best_val_loss'} <= set(so)
  # a second trainer resumes from it and produces the identical next step
  opt2 = _opts(load_model=True, checkpoints_dir=str(tmp_path), **kw)
  tr2 = get_trainer(opt2)(opt2, 'train')
  tr2.load()
  assert tr2.steps_taken == 1 and tr2.start_epoch == 1
  xd2 = omodel.synthetic_batch(1, 32, 64, seed=2)
  a, b = tr.step(xd2), tr2.step(xd2)
  assert abs(a - b) <= 1e-6 * max(abs(a), 1e-6)


def test_bf16_step_tracks_fp32():
  kw = dict(ngf=8, ndf=8, n_blocks_global=2)
  tr32, ora, _ = _paired(kw)
  opt16 = _opts(compute_dtype='bf16', **kw)
  tr16 = get_trainer(opt16)(opt16, 'train')
  tr16.model.netG.load_state_dict(tr32.model.netG.state_dict())
  tr16.model.netD.load_state_dict(tr32.model.netD.state_dict())
  xd = omodel.synthetic_batch(2, 32, 64, seed=11)
  tr32.step(xd)
  tr16.step(xd)
  for k in omodel.LOSS_NAMES:
    a, b = tr16.last_losses[k], tr32.last_losses[k]
    assert abs(a - b) <= 5e-2 * max(abs(b), 1e-3), (k, a, b)


def test_unsupported_flags_fail_loudly():
  with pytest.raises(NotImplementedError):
    get_trainer(_opts(no_generator_binarization=False))(_opts(no_generator_binarization=False), 'train')
  with pytest.raises(NotImplementedError):
    get_trainer(_opts(pool_size=5))(_opts(pool_size=5), 'train')
