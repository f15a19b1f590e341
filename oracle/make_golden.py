"""Generate tests/golden/*.npz from the REAL reference (build container only).

Run:  python -m oracle.make_golden

Every fixture holds data only: configuration scalars, seeded inputs, reference
outputs.  Large weights are not stored: they are regenerated from the torch
seed by `oracle.ctu_cpu.nets.init_*`, which `oracle/check_against_reference.py`
proves bit-identical to the reference's define_G/define_D under that seed.
"""
import copy
import os

import numpy as np
import torch

from oracle import _refbridge
from oracle.ctu_cpu import nets, model as omodel

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
SEED = 1234


def _np(t):
  return t.detach().cpu().numpy()


def _clone(xd):
  return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in xd.items()}


def _grad_record(named_params, full_for):
  """Per-parameter grad norms for every tensor; full grads only for `full_for` keys."""
  rec, norms, keys = {}, [], []
  for k, p in named_params:
    keys.append(k)
    norms.append(float(p.grad.double().norm()))
    if any(k.startswith(f) for f in full_for):
      rec['g:' + k] = _np(p.grad)
  rec['gradnorms'] = np.array(norms)
  rec['keys'] = np.array(keys)
  return rec


def golden_networks(networks):
  """G2: netG (global, local), netD (all 10 maps), VGG (5 maps); fwd + input grads."""
  out = {}
  # ---- global generator, ngf=8, 2 blocks ----
  torch.manual_seed(SEED)
  g = networks.define_G(39, 3, 8, 'global', 4, 2, 1, 3, 'instance', gpu_ids=[],
                        binarize_generator=False)
  x = (torch.rand(2, 39, 32, 64, generator=torch.Generator().manual_seed(7)) - 0.5).requires_grad_(True)
  y = g(x)
  r = torch.rand(y.shape, generator=torch.Generator().manual_seed(8)) - 0.5
  (y * r).sum().backward()
  out['netG_global_ngf8'] = dict(
      x=_np(x), y=_np(y), r=_np(r), dx=_np(x.grad), seed=np.int64(SEED),
      **_grad_record(g.named_parameters(), ('model.1.', 'model.4.', 'model.16.conv_block.5.',
                                            'model.24.', 'model.31.')))
  # ---- local enhancer, ngf=4 ----
  torch.manual_seed(SEED)
  g = networks.define_G(39, 3, 4, 'local', 4, 2, 1, 3, 'instance', gpu_ids=[],
                        binarize_generator=False)
  x = (torch.rand(1, 39, 64, 128, generator=torch.Generator().manual_seed(9)) - 0.5).requires_grad_(True)
  y = g(x)
  r = torch.rand(y.shape, generator=torch.Generator().manual_seed(10)) - 0.5
  (y * r).sum().backward()
  out['netG_local_ngf4'] = dict(
      x=_np(x), y=_np(y), r=_np(r), dx=_np(x.grad), seed=np.int64(SEED),
      **_grad_record(g.named_parameters(), ('model1_1.', 'model1_2.0.conv_block.1.',
                                            'model1_2.3.', 'model1_2.7.', 'model.1.')))
  # ---- discriminator, ndf=8 ----
  torch.manual_seed(SEED)
  d = networks.define_D(39, 8, 3, 'instance', False, 2, True, gpu_ids=[])
  x = (torch.rand(2, 39, 32, 64, generator=torch.Generator().manual_seed(11)) - 0.5).requires_grad_(True)
  feats = d(x)
  total = 0
  for i, scale in enumerate(feats):
    for j, f in enumerate(scale):
      total = total + (f * (0.1 + 0.05 * (i * 5 + j))).sum()
  total.backward()
  rec = dict(x=_np(x), dx=_np(x.grad))
  for i, scale in enumerate(feats):
    for j, f in enumerate(scale):
      rec['f:%d:%d' % (i, j)] = _np(f)
  rec['seed'] = np.int64(SEED)
  rec.update(_grad_record(d.named_parameters(), ('scale0_layer0.', 'scale1_layer2.', 'scale0_layer4.')))
  out['netD_ndf8'] = rec
  # ---- VGG19 slices on the seeded weights (weights regenerated from seed 20) ----
  vgg = networks.Vgg19()
  x = (torch.rand(1, 3, 32, 64, generator=torch.Generator().manual_seed(12)) - 0.5).requires_grad_(True)
  maps = vgg(x)
  total = 0
  for k, m in enumerate(maps):
    total = total + nets.VGG_LOSS_WEIGHTS[k] * m.abs().mean()
  total.backward()
  rec = dict(x=_np(x), dx=_np(x.grad), vgg_seed=np.int64(20))
  for k, m in enumerate(maps):
    rec['m:%d' % k] = _np(m)
  out['vgg19_seed20'] = rec
  return out


def golden_steps(RefTrainer):
  """G3: three consecutive step()s: six losses, returned scalar, parameter / grad norms."""
  out = {}
  cases = {
      'step_global_ngf8': (omodel.default_opt(ngf=8, ndf=8, n_blocks_global=2), 32, 64, 2, 3),
      'step_local_ngf4': (omodel.default_opt(netG='local', ngf=4, ndf=8, n_blocks_global=2,
                                             use_compressed=False), 64, 128, 1, 3),
      'step_global_ngf64_full': (omodel.default_opt(), 32, 64, 1, 2),
  }
  for name, (opt, h, w, b, steps) in cases.items():
    torch.manual_seed(SEED)
    tr = RefTrainer(copy.deepcopy(opt), 'train')
    rec = dict(height=np.int64(h), width=np.int64(w), batch=np.int64(b), steps=np.int64(steps),
               seed=np.int64(SEED), opt_netG=np.array(opt.netG), opt_ngf=np.int64(opt.ngf),
               opt_ndf=np.int64(opt.ndf), opt_n_blocks_global=np.int64(opt.n_blocks_global))
    for s in range(steps):
      xd = omodel.synthetic_batch(b, h, w, seed=100 + s, num_labels=opt.num_labels)
      # losses of this step before the update, from an extra forward (same numbers step() prints)
      tr.train()
      L = tr.model(_clone(xd), tr.opt, mode='get_train_loss')
      rec['losses:%d' % s] = np.array([float(v) for v in L], dtype=np.float64)
      ret = tr.step(_clone(xd))
      rec['ret:%d' % s] = np.float64(ret)
      pn = {k: float(v.double().norm()) for k, v in tr.model.netG.state_dict().items()}
      rec['Gnorm:%d' % s] = np.array(list(pn.values()))
      dn = {k: float(v.double().norm()) for k, v in tr.model.netD.state_dict().items()}
      rec['Dnorm:%d' % s] = np.array(list(dn.values()))
      gn = [0.0 if p.grad is None else float(p.grad.double().norm())
            for _, p in tr.model.netG.named_parameters()]
      rec['Ggradnorm:%d' % s] = np.array(gn)
      gn = [0.0 if p.grad is None else float(p.grad.double().norm())
            for _, p in tr.model.netD.named_parameters()]
      rec['Dgradnorm:%d' % s] = np.array(gn)
    rec['Gkeys'] = np.array(list(tr.model.netG.state_dict().keys()))
    rec['Dkeys'] = np.array(list(tr.model.netD.state_dict().keys()))
    xd = omodel.synthetic_batch(b, h, w, seed=999, num_labels=opt.num_labels)
    rec['get_img'] = _np(tr.get_img(_clone(xd)))
    rec['get_eval_loss'] = np.float64(tr.get_eval_loss(_clone(xd)))
    out[name] = rec
  return out


def golden_preprocess(RefModel):
  """G4/G5: one-hot + edge on a crop of the bundled Cityscapes maps; tensor2im truncation."""
  from PIL import Image
  d = os.path.join(_refbridge.REFERENCE_ROOT,
                   'datasets/cityscapes_test_CVPR20_1024/gtFine/val/frankfurt')
  stem = 'frankfurt_000000_005898_gtFine_'
  lab = np.array(Image.open(os.path.join(d, stem + 'labelIds.png')))[180:244, 400:528]
  ins = np.array(Image.open(os.path.join(d, stem + 'instanceIds.png')))[180:244, 400:528]
  opt = omodel.default_opt(is_train=False)
  opt.checkpoints_dir = '/nonexistent'
  # the reference's preprocess only needs `self.opt`, FloatTensor/ByteTensor: build a bare instance
  m = RefModel.__new__(RefModel)
  torch.nn.Module.__init__(m)
  m.opt = opt
  m.FloatTensor, m.ByteTensor = torch.FloatTensor, torch.ByteTensor
  label_t = torch.tensor(lab.astype(np.float32))[None, None]
  label_t[label_t == 255] = opt.num_labels
  inst_t = torch.tensor(ins.astype(np.int64))[None, None]
  img = torch.rand(1, 3, 64, 128, generator=torch.Generator().manual_seed(3)) - 0.5
  res = m.preprocess({'label': label_t.clone(), 'instance': inst_t.clone(), 'image': img})
  from ctu.utils.misc import tensor2im
  probe = torch.linspace(-0.75, 0.75, 3 * 16 * 32).reshape(1, 3, 16, 32)
  return {'preprocess_cityscapes_crop': dict(
      label=lab.astype(np.uint8), instance=ins.astype(np.int32),
      input_label=_np(res['input_label']).astype(np.uint8),
      t2i_in=_np(probe), t2i_out=tensor2im(probe, opt))}


def main():
  torch.set_num_threads(8)
  torch.use_deterministic_algorithms(True)
  os.makedirs(OUT, exist_ok=True)
  networks, RefModel, RefTrainer = _refbridge.import_reference(nets.init_vgg19())
  records = {}
  records.update(golden_networks(networks))
  records.update(golden_preprocess(RefModel))
  records.update(golden_steps(RefTrainer))
  for name, rec in records.items():
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **rec)
    print('%-32s %8.1f KB' % (name, os.path.getsize(path) / 1024.0))


if __name__ == '__main__':
  main()
