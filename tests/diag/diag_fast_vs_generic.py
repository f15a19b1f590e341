"""Per-layer cosine of the bf16 weight gradients: fast kernels (mode 1) vs generic kernels (mode 0),
same weights and batch (ngf=64 generator at 128x256, batch 2)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import jpdse_hip
from jpdse_hip import lib
from ctu.trainers import get_trainer
from oracle.ctu_cpu import model as omodel

modes = [int(a) for a in sys.argv[1:]] or [1, 0]
torch.manual_seed(1234)
opt = omodel.default_opt(gpu_ids=[0], print_losses=False, compute_dtype='bf16')
xd = omodel.synthetic_batch(2, 128, 256, seed=21)
base = get_trainer(opt)(opt, 'train')
sdG, sdD = base.model.netG.state_dict(), base.model.netD.state_dict()
res = {}
for m in modes:
  jpdse_hip.set_dev_mode(m)
  tr = get_trainer(opt)(opt, 'train')
  tr.model.netG.load_state_dict(sdG); tr.model.netD.load_state_dict(sdD)
  tr.step(xd)
  res[m] = ({k: p.grad.detach().cpu().double().flatten() for k, p in list(tr.model.netG.named_parameters()) + [('D.' + k, p) for k, p in tr.model.netD.named_parameters()]
             if k.endswith('.weight')}, dict(tr.last_losses))
jpdse_hip.set_dev_mode(1)
opt32 = omodel.default_opt(gpu_ids=[0], print_losses=False, compute_dtype='fp32')
tr = get_trainer(opt32)(opt32, 'train')
tr.model.netG.load_state_dict(sdG); tr.model.netD.load_state_dict(sdD)
tr.step(xd)
ref = {k: p.grad.detach().cpu().double().flatten() for k, p in list(tr.model.netG.named_parameters()) + [('D.' + k, p) for k, p in tr.model.netD.named_parameters()]
       if k.endswith('.weight')}
cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30))
a, b = res[modes[0]][0], res[modes[1]][0]
print('losses', res[modes[0]][1]); print('losses', res[modes[1]][1])
for k in a:
  print('%-34s cos %.5f  norm ratio %.4f   vs fp32: mode%d %.5f  mode%d %.5f' % (k, cos(a[k], b[k]), float(a[k].norm() / b[k].norm().clamp_min(1e-30)), modes[0], cos(a[k], ref[k]), modes[1], cos(b[k], ref[k])))
