"""Diagnostic: bf16 gradient agreement -- fast kernels on vs off vs fp32 (conditioning or bug?)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle.ctu_cpu import model as omodel
from ctu.trainers import get_trainer
import jpdse_hip
from jpdse_hip import lib
import test_hip_step as T

def grads(dtype, fast):
  jpdse_hip.set_dev_mode(fast)
  opt = T._opts(compute_dtype=dtype)
  tr = get_trainer(opt)(opt, 'train')
  tr.model.netG.load_state_dict({k: v.detach() for k, v in ora.G.items()})
  tr.model.netD.load_state_dict({k: v.detach() for k, v in ora.D.items()})
  tr.step(xd)
  g = {k: p.grad.detach().cpu().double().flatten().clone() for k, p in tr.model.netG.named_parameters() if k.endswith('.weight')}
  L = dict(tr.last_losses)
  del tr
  torch.cuda.empty_cache()
  return g, L

torch.manual_seed(1234)
ora = omodel.OracleTrainer(omodel.default_opt())
xd = omodel.synthetic_batch(2, 128, 256, seed=21)
gref, _ = ora.grads_in_dtype(xd, torch.float32)
gref = {k: v.double().flatten() for k, v in gref.items() if k.endswith('.weight')}
from oracle.ctu_cpu import nets as onets
onets.storage_bf16(True)
gemu, _ = ora.grads_in_dtype(xd, torch.float32)
onets.storage_bf16(False)
gemu = {k: v.double().flatten() for k, v in gemu.items() if k.endswith('.weight')}
cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()))
g32, L32 = grads('fp32', 1)
g16f, L16f = grads('bf16', 1)
g16g, L16g = grads('bf16', 0)
print('losses fp32', {k: round(v, 5) for k, v in L32.items()})
print('losses bf16 fast', {k: round(v, 5) for k, v in L16f.items()})
print('losses bf16 generic', {k: round(v, 5) for k, v in L16g.items()})
print('%-34s %9s %9s %9s %9s %9s %9s' % ('param', 'fp32~ref', 'bf16f~ref', 'bf16g~ref', 'bf16f~g', 'emu~ref', 'bf16f~emu'))
for k in gref:
  print('%-34s %9.5f %9.5f %9.5f %9.5f %9.5f %9.5f' % (k, cos(g32[k], gref[k]), cos(g16f[k], gref[k]), cos(g16g[k], gref[k]), cos(g16f[k], g16g[k]), cos(gemu[k], gref[k]), cos(g16f[k], gemu[k])))
