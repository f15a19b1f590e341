"""which kernel family moved the deep-layer bf16-vs-fp32 weight-gradient cosine of the LocalEnhancer in-situ test
(tests/test_hip_step.py::test_bf16_local_enhancer_full_width_in_situ)?  Same weights and batch, fp32 HIP path as the yardstick,
bf16 path under developer modes 1 (shipped selection), 29 (no row-streaming kernels), 32 (no fused InstanceNorm moments).
Usage: python scripts/diag_cos_modes.py [modes ...]"""
import sys, os, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch
import jpdse_hip
from ctu.trainers import get_trainer
from oracle.ctu_cpu import model as omodel

modes = [int(a) for a in sys.argv[1:]] or [1, 29, 32]
kw = dict(netG='local', ngf=32)
xd = omodel.synthetic_batch(1, 128, 256, seed=33)
opt32 = omodel.default_opt(gpu_ids=[0], print_losses=False, **kw)
torch.manual_seed(77)
tr32 = get_trainer(opt32)(opt32, 'train')
sdG = {k: v.detach().clone() for k, v in tr32.model.netG.state_dict().items()}
sdD = {k: v.detach().clone() for k, v in tr32.model.netD.state_dict().items()}
tr32.step(xd)
g32 = {k: p.grad.detach().cpu().double().flatten() for k, p in tr32.model.netG.named_parameters() if k.endswith('.weight')}
del tr32
cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30))
for m in modes:
  with (jpdse_hip.dev_mode(m) if m != 1 else contextlib.nullcontext()):
    opt16 = omodel.default_opt(gpu_ids=[0], print_losses=False, compute_dtype='bf16', **kw)
    tr16 = get_trainer(opt16)(opt16, 'train')
    tr16.model.netG.load_state_dict(sdG)
    tr16.model.netD.load_state_dict(sdD)
    tr16.step(xd)
    torch.cuda.synchronize()
    cs = sorted((cos(p.grad.detach().cpu().double().flatten(), g32[k]), k) for k, p in tr16.model.netG.named_parameters()
                if k.endswith('.weight'))
    print('mode %2d: lowest cosines %s' % (m, ', '.join('%s %.4f' % (k, c) for c, k in cs[:4])))
    del tr16
