"""Diagnostic: is the HIP-vs-oracle gradient gap a bug or conditioning?  Compare both with an fp64 run."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from oracle.ctu_cpu import model as omodel, nets
from ctu.trainers import get_trainer

def oracle_grads(ora, xd, dt):
  """loss_G grads of the oracle's CURRENT weights in dtype dt."""
  opt = ora.opt
  G = {k: v.detach().to(dt).requires_grad_(True) for k, v in ora.G.items()}
  D = {k: v.detach().to(dt) for k, v in ora.D.items()}
  V = {k: v.to(dt) for k, v in ora.vgg.items()}
  lab = omodel.preprocess(xd, opt).to(dt)
  real = xd['image'].to(dt)
  fake = nets.generator(G, torch.cat((lab, real), 1), ora.cfg)
  pf = nets.multiscale_d(D, torch.cat((lab, fake), 1), opt.num_D, opt.n_layers_D)
  pr = nets.multiscale_d(D, torch.cat((lab, real), 1), opt.num_D, opt.n_layers_D)
  import torch.nn.functional as F
  loss = nets.gan_loss(pf, True)
  for i in range(opt.num_D):
    for j in range(len(pf[i]) - 1):
      loss = loss + opt.lambda_feat * (1.0 / opt.num_D) * F.l1_loss(pf[i][j], pr[i][j].detach())
  loss = loss + opt.lambda_feat * nets.vgg_loss(V, fake, real) + opt.lambda_distortion * F.l1_loss(fake, real)
  loss.backward()
  return {k: v.grad for k, v in G.items()}

kw = dict(ngf=8, ndf=8, n_blocks_global=2)
torch.manual_seed(1234)
ora = omodel.OracleTrainer(omodel.default_opt(**kw))
opt = omodel.default_opt(gpu_ids=[0], print_losses=False, **kw)
tr = get_trainer(opt)(opt, 'train')
tr.model.netG.load_state_dict({k: v.detach() for k, v in ora.G.items()})
tr.model.netD.load_state_dict({k: v.detach() for k, v in ora.D.items()})
xd = omodel.synthetic_batch(2, 128, 256, seed=100)
g64 = oracle_grads(ora, xd, torch.float64)
g32 = oracle_grads(ora, xd, torch.float32)
tr.step(xd)
for k, p in tr.model.netG.named_parameters():
  if not k.endswith('.weight'): continue
  t = g64[k]
  e_hip = ((p.grad.cpu().double() - t).norm() / t.norm()).item()
  e_t32 = ((g32[k].double() - t).norm() / t.norm()).item()
  print('%-32s l2-rel vs fp64:  HIP %.2e   torch-fp32 %.2e' % (k, e_hip, e_t32))
