import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle.ctu_cpu import model as omodel, nets
from ctu.trainers import get_trainer
from hip_util import rel_err
import test_hip_step as T

kw = dict(ngf=8, ndf=8, n_blocks_global=2)
tr, ora, opt = T._paired(kw)
xq = omodel.synthetic_batch(2, 32, 64, seed=999)
print('init   get_img err', rel_err(tr.get_img(xq).cpu(), ora.get_img(xq)))
for s in range(2):
  xd = omodel.synthetic_batch(2, 32, 64, seed=100 + s)
  tr.step(xd); ora.step(xd)
  print('step', s, 'get_img err (own trajectories)', rel_err(tr.get_img(xq).cpu(), ora.get_img(xq)))
T._sync_from_oracle(tr, ora)
w_err = max(rel_err(v.cpu(), ora.G[k].detach()) for k, v in tr.model.netG.state_dict().items())
print('after sync: weight err', w_err)
print('after sync get_img err', rel_err(tr.get_img(xq).cpu(), ora.get_img(xq)))
lab = omodel.preprocess(xq, opt)
x = torch.cat((lab, xq['image']), 1)
print('after sync netG(x) NCHW err', rel_err(tr.model.netG(x.cuda()).cpu(), ora.get_img(xq)))
# biases: are the dead biases large now?
for k, v in ora.G.items():
  if k.endswith('.bias'): print(k, float(v.abs().max())); break
# what if oracle zeroes its dead biases?
G0 = {k: (torch.zeros_like(v) if (k.endswith('.bias') and k != 'model.31.bias') else v.detach()) for k, v in ora.G.items()}
y0 = nets.generator(G0, x, ora.cfg)
print('oracle with zeroed dead biases vs oracle', rel_err(y0.detach(), ora.get_img(xq)))
print('HIP vs oracle-with-zeroed-dead-biases', rel_err(tr.get_img(xq).cpu(), y0.detach()))
