"""CPU: the oracle (torch-CPU restatement) against golden vectors produced by the real
reference (oracle/make_golden.py).  These pin the checker used by every GPU parity test."""
import os

import numpy as np
import torch

from oracle.ctu_cpu import nets, model as omodel

RTOL = 2e-5


def _load(golden_dir, name):
  return np.load(os.path.join(golden_dir, name + '.npz'))


def _close(a, b, rtol=RTOL):
  a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
  scale = max(np.abs(b).max(), 1e-12)
  assert np.abs(a - b).max() <= rtol * scale, (np.abs(a - b).max(), scale)


def _leaf(sd):
  return {k: v.clone().requires_grad_(True) for k, v in sd.items()}


def test_global_generator_fwd_bwd(golden_dir):
  g = _load(golden_dir, 'netG_global_ngf8')
  torch.manual_seed(int(g['seed']))
  cfg = dict(netG='global', ngf=8, n_downsample_global=4, n_blocks_global=2,
             n_local_enhancers=1, n_blocks_local=3)
  sd = _leaf(nets.init_generator(cfg, 39, 3))
  assert list(sd.keys()) == list(g['keys'])
  x = torch.tensor(g['x']).requires_grad_(True)
  y = nets.generator(sd, x, cfg)
  _close(y.detach().numpy(), g['y'])
  (y * torch.tensor(g['r'])).sum().backward()
  _close(x.grad.numpy(), g['dx'], 1e-4)
  norms = np.array([float(sd[k].grad.double().norm()) for k in g['keys']])
  live = np.array([not (k.endswith('.bias') and k != 'model.%d.bias' % nets.global_layout(4, 2)[4])
                   for k in g['keys']])
  np.testing.assert_allclose(norms[live], g['gradnorms'][live], rtol=1e-4)
  for k in g.files:
    if k.startswith('g:') and k.endswith('.weight'):
      _close(sd[k[2:]].grad.numpy(), g[k], 1e-4)


def test_local_enhancer_fwd_bwd(golden_dir):
  g = _load(golden_dir, 'netG_local_ngf4')
  torch.manual_seed(int(g['seed']))
  cfg = dict(netG='local', ngf=4, n_downsample_global=4, n_blocks_global=2,
             n_local_enhancers=1, n_blocks_local=3)
  sd = _leaf(nets.init_generator(cfg, 39, 3))
  assert list(sd.keys()) == list(g['keys'])
  x = torch.tensor(g['x']).requires_grad_(True)
  y = nets.generator(sd, x, cfg)
  _close(y.detach().numpy(), g['y'])
  (y * torch.tensor(g['r'])).sum().backward()
  _close(x.grad.numpy(), g['dx'], 1e-4)
  for k in g.files:
    if k.startswith('g:') and k.endswith('.weight'):
      _close(sd[k[2:]].grad.numpy(), g[k], 1e-4)


def test_discriminator_features(golden_dir):
  g = _load(golden_dir, 'netD_ndf8')
  torch.manual_seed(int(g['seed']))
  sd = _leaf(nets.init_discriminator(39, 8, 3, 2))
  assert list(sd.keys()) == list(g['keys'])
  x = torch.tensor(g['x']).requires_grad_(True)
  feats = nets.multiscale_d(sd, x, 2, 3)
  total = 0
  for i, scale in enumerate(feats):
    assert len(scale) == 5
    for j, f in enumerate(scale):
      _close(f.detach().numpy(), g['f:%d:%d' % (i, j)])
      total = total + (f * (0.1 + 0.05 * (i * 5 + j))).sum()
  total.backward()
  _close(x.grad.numpy(), g['dx'], 1e-4)
  for k in g.files:
    if k.startswith('g:') and k.endswith('.weight'):
      _close(sd[k[2:]].grad.numpy(), g[k], 1e-4)


def test_vgg_feature_pyramid(golden_dir):
  g = _load(golden_dir, 'vgg19_seed20')
  sd = nets.init_vgg19(int(g['vgg_seed']))
  x = torch.tensor(g['x']).requires_grad_(True)
  maps = nets.vgg19_features(sd, x)
  assert [m.shape[1] for m in maps] == [64, 128, 256, 512, 512]
  total = 0
  for k, m in enumerate(maps):
    _close(m.detach().numpy(), g['m:%d' % k])
    total = total + nets.VGG_LOSS_WEIGHTS[k] * m.abs().mean()
  total.backward()
  _close(x.grad.numpy(), g['dx'], 1e-4)


def test_preprocess_and_tensor2im_integer_exact(golden_dir):
  g = _load(golden_dir, 'preprocess_cityscapes_crop')
  opt = omodel.default_opt()
  lab = torch.tensor(g['label'].astype(np.float32))[None, None]
  lab[lab == 255] = opt.num_labels
  ins = torch.tensor(g['instance'].astype(np.int64))[None, None]
  out = omodel.preprocess({'label': lab, 'instance': ins}, opt)
  assert out.shape == (1, 36, 64, 128)
  assert np.array_equal(out.numpy().astype(np.uint8), g['input_label'])
  assert np.array_equal(omodel.tensor2im(torch.tensor(g['t2i_in']), opt), g['t2i_out'])


def _run_steps(golden_dir, name):
  g = _load(golden_dir, name)
  opt = omodel.default_opt(netG=str(g['opt_netG']), ngf=int(g['opt_ngf']), ndf=int(g['opt_ndf']),
                           n_blocks_global=int(g['opt_n_blocks_global']))
  torch.manual_seed(int(g['seed']))
  tr = omodel.OracleTrainer(opt)
  assert list(tr.G.keys()) == list(g['Gkeys']) and list(tr.D.keys()) == list(g['Dkeys'])
  b, h, w = int(g['batch']), int(g['height']), int(g['width'])
  for s in range(int(g['steps'])):
    xd = omodel.synthetic_batch(b, h, w, seed=100 + s, num_labels=opt.num_labels)
    ret = tr.step(xd)
    np.testing.assert_allclose(list(tr.last_losses.values()), g['losses:%d' % s], rtol=1e-4)
    np.testing.assert_allclose(ret, float(g['ret:%d' % s]), rtol=1e-4)
    wmask = np.array([k.endswith('.weight') for k in g['Gkeys']])
    norms = np.array([float(v.detach().double().norm()) for v in tr.G.values()])
    np.testing.assert_allclose(norms[wmask], g['Gnorm:%d' % s][wmask], rtol=1e-4)
  xd = omodel.synthetic_batch(b, h, w, seed=999, num_labels=opt.num_labels)
  _close(tr.get_img(xd).numpy(), g['get_img'], 1e-3)
  np.testing.assert_allclose(tr.get_eval_loss(xd), float(g['get_eval_loss']), rtol=1e-3)


def test_three_train_steps_global(golden_dir):
  _run_steps(golden_dir, 'step_global_ngf8')


def test_three_train_steps_local(golden_dir):
  _run_steps(golden_dir, 'step_local_ngf4')


def test_two_train_steps_full_width(golden_dir):
  _run_steps(golden_dir, 'step_global_ngf64_full')


def test_bf16_storage_emulation_is_opt_in():
  """nets.storage_bf16(False) (the default) must leave the pinned fp32 oracle untouched; when on,
  every produced activation is bf16-representable and gradients still flow to the fp32 masters."""
  cfg = dict(netG='global', ngf=8, n_downsample_global=2, n_blocks_global=1, n_local_enhancers=1, n_blocks_local=3)
  torch.manual_seed(3)
  sd = _leaf(nets.init_generator(cfg, 39, 3))
  x = torch.rand(1, 39, 16, 32) - 0.5
  y0 = nets.generator(sd, x, cfg)
  nets.storage_bf16(True)
  try:
    y1 = nets.generator(sd, x, cfg)
    y1.sum().backward()
  finally:
    nets.storage_bf16(False)
  y2 = nets.generator(sd, x, cfg)
  assert torch.equal(y0, y2)
  assert torch.equal(y1, y1.to(torch.bfloat16).float()) and not torch.equal(y0, y1)
  assert all(v.grad is not None and v.grad.dtype == torch.float32 for k, v in sd.items() if k.endswith('.weight'))
