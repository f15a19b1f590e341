"""Pin the oracle against the real reference (build container only; TEST INFRASTRUCTURE).

Run:  python -m oracle.check_against_reference [--quick]

For each configuration it builds the reference `Pix2PixHDTrainer` and the oracle
`OracleTrainer` under the same torch seed and checks
  * define_G / define_D state dicts are bit-identical (RNG-order restatement),
  * generator output, the six losses, every parameter gradient, and the parameters
    after each of three `step()`s agree,
  * get_img / get_eval_loss agree.
Prints one PASS/FAIL line per check; exit code 1 on any failure.
"""
import argparse
import copy
import sys

import torch

from oracle import _refbridge
from oracle.ctu_cpu import nets, model as omodel


def _maxrel(a, b):
  d = (a - b).abs().max().item()
  s = max(b.abs().max().item(), 1e-12)
  return d / s


def check_config(name, opt, height, width, batch, steps=3, tol=2e-5):
  ok = True
  def report(what, good, detail=''):
    nonlocal ok
    ok &= bool(good)
    print('[%s] %-34s %s %s' % (name, what, 'PASS' if good else 'FAIL', detail))

  vgg_sd = nets.init_vgg19()
  networks, RefModel, RefTrainer = _refbridge.import_reference(vgg_sd)
  torch.manual_seed(1234)
  ref = RefTrainer(copy.deepcopy(opt), 'train')
  torch.manual_seed(1234)
  ora = omodel.OracleTrainer(copy.deepcopy(opt), sd_vgg=vgg_sd)

  # --- init parity (bit exact) ---
  rg, rd = ref.model.netG.state_dict(), ref.model.netD.state_dict()
  report('G keys/shapes', list(rg.keys()) == list(ora.G.keys())
         and all(rg[k].shape == ora.G[k].shape for k in rg))
  report('D keys/shapes', list(rd.keys()) == list(ora.D.keys())
         and all(rd[k].shape == ora.D[k].shape for k in rd))
  report('G init bit-exact', all(torch.equal(rg[k], ora.G[k].detach()) for k in rg))
  report('D init bit-exact', all(torch.equal(rd[k], ora.D[k].detach()) for k in rd))

  for s in range(steps):
    xd = omodel.synthetic_batch(batch, height, width, seed=100 + s, num_labels=opt.num_labels)
    ref_in = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in xd.items()}
    # reference losses + grads via its own step (prints the six losses)
    r_ret = ref.step(ref_in)
    o_ret = ora.step(xd, keep_grads=True)
    report('step %d return (G_Distortion)' % s, abs(r_ret - o_ret) <= tol * max(abs(r_ret), 1e-6),
           '%.7f vs %.7f' % (r_ret, o_ret))
    # parameters after the step: skip biases in front of affine-less InstanceNorm
    # (true gradient 0; Adam turns rounding noise into +-lr: SURVEY.md §7 "Hard parts")
    worst = 0.0
    for k, v in ref.model.netG.state_dict().items():
      if k.endswith('.bias') and not _live_bias_G(k, opt):
        continue
      worst = max(worst, _maxrel(ora.G[k].detach(), v))
    report('step %d G params' % s, worst < 5e-3, 'max rel %.2e' % worst)
    worst = 0.0
    for k, v in ref.model.netD.state_dict().items():
      if k.endswith('.bias') and not _live_bias_D(k, opt):
        continue
      worst = max(worst, _maxrel(ora.D[k].detach(), v))
    report('step %d D params' % s, worst < 5e-3, 'max rel %.2e' % worst)

  # --- forward-only parity on a fresh batch, plus grads from one loss graph ---
  xd = omodel.synthetic_batch(batch, height, width, seed=999, num_labels=opt.num_labels)
  ref_img = ref.get_img({k: (v.clone() if torch.is_tensor(v) else v) for k, v in xd.items()})
  ora_img = ora.get_img(xd)
  report('get_img', _maxrel(ora_img, ref_img) < tol, 'max rel %.2e' % _maxrel(ora_img, ref_img))
  r_ev = ref.get_eval_loss({k: (v.clone() if torch.is_tensor(v) else v) for k, v in xd.items()})
  o_ev = ora.get_eval_loss(xd)
  report('get_eval_loss (uint8 scale)', abs(r_ev - o_ev) < 1e-3 * max(r_ev, 1.0), '%.5f vs %.5f' % (r_ev, o_ev))

  ref.train()
  r_losses = ref.model({k: (v.clone() if torch.is_tensor(v) else v) for k, v in xd.items()},
                       ref.opt, mode='get_train_loss')
  o_losses = ora.train_losses(xd)
  for nm, a, b in zip(omodel.LOSS_NAMES, r_losses, o_losses):
    report('loss ' + nm, abs(float(a) - float(b)) <= tol * max(abs(float(a)), 1e-6),
           '%.7f vs %.7f' % (float(a), float(b)))
  ref.optimizer_G.zero_grad(); ref.optimizer_D.zero_grad()
  for v in list(ora.G.values()) + list(ora.D.values()):
    v.grad = None
  (r_losses[0] + 10 * r_losses[1] + 10 * r_losses[2] + 10 * r_losses[3]).backward()
  (o_losses[0] + 10 * o_losses[1] + 10 * o_losses[2] + 10 * o_losses[3]).backward()
  worst, worst_dead = 0.0, 0.0
  for (k, p) in ref.model.netG.named_parameters():
    g = ora.G[k].grad
    if k.endswith('.bias') and not _live_bias_G(k, opt):
      worst_dead = max(worst_dead, (g - p.grad).abs().max().item())
      continue
    worst = max(worst, _maxrel(g, p.grad))
  report('G grads (live params)', worst < 1e-3, 'max rel %.2e; dead-bias abs %.2e' % (worst, worst_dead))
  return ok


def _live_bias_G(key, opt):
  """Only the last 7x7 conv's bias is not followed by an affine-less InstanceNorm."""
  if opt.netG == 'global':
    last = nets.global_layout(opt.n_downsample_global, opt.n_blocks_global)[4]
    return key == 'model.%d.bias' % last
  return key == 'model%d_2.%d.bias' % (opt.n_local_enhancers, opt.n_blocks_local + 4)


def _live_bias_D(key, opt):
  return ('_layer0.' in key) or ('_layer%d.' % (opt.n_layers_D + 1) in key)


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument('--quick', action='store_true', help='small widths only')
  args = ap.parse_args()
  torch.set_num_threads(8)
  results = []
  results.append(check_config('global-ngf8', omodel.default_opt(ngf=8, ndf=8, n_blocks_global=2),
                              32, 64, 2))
  results.append(check_config('local-ngf4', omodel.default_opt(netG='local', ngf=4, ndf=8,
                                                               n_blocks_global=2), 64, 128, 1))
  results.append(check_config('global-mse',
                              omodel.default_opt(ngf=8, ndf=8, n_blocks_global=1,
                                                 distortion_loss_fn='mse'), 32, 64, 1))
  if not args.quick:
    results.append(check_config('global-ngf64-full', omodel.default_opt(), 32, 64, 1, steps=2))
  print('ALL PASS' if all(results) else 'SOME FAILED')
  sys.exit(0 if all(results) else 1)


if __name__ == '__main__':
  main()
