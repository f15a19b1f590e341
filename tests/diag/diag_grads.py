"""Diagnostic: per-parameter gradient error of one HIP train step vs the oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle.ctu_cpu import model as omodel
from ctu.trainers import get_trainer

def run(kw, b, h, w):
  torch.manual_seed(1234)
  ora = omodel.OracleTrainer(omodel.default_opt(**kw))
  opt = omodel.default_opt(gpu_ids=[0], print_losses=False, **kw)
  tr = get_trainer(opt)(opt, 'train')
  tr.model.netG.load_state_dict({k: v.detach() for k, v in ora.G.items()})
  tr.model.netD.load_state_dict({k: v.detach() for k, v in ora.D.items()})
  xd = omodel.synthetic_batch(b, h, w, seed=100)
  tr.step(xd); ora.step(xd, keep_grads=True)
  print(kw, {k: (round(tr.last_losses[k], 6), round(ora.last_losses[k], 6)) for k in omodel.LOSS_NAMES})
  for net, ref, tag in ((tr.model.netG, ora.grads_G, 'G'), (tr.model.netD, ora.grads_D, 'D')):
    for k, p in net.named_parameters():
      if not k.endswith('.weight'): continue
      a, r = p.grad.cpu().double(), ref[k].double()
      print('%s %-32s max-rel %.2e  l2-rel %.2e  |ref|max %.2e' % (tag, k, ((a-r).abs().max()/r.abs().max()).item(), ((a-r).norm()/r.norm()).item(), r.abs().max().item()))

run(dict(ngf=8, ndf=8, n_blocks_global=2), 2, 32, 64)
run(dict(ngf=8, ndf=8, n_blocks_global=2), 2, 128, 256)
