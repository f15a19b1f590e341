"""Diagnostic: forward rounding error of the HIP fp32 path vs torch-fp32, both against fp64."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch, torch.nn.functional as F
from jpdse_hip import F32, PAD_REFLECT, ACT_NONE, ACT_RELU
from jpdse_hip.layers import HipConv2d, InstNormAct
from hip_util import DEV, to_act, to_nchw
from oracle.ctu_cpu import nets
from ctu.models.pix2pixHD_networks import networks

def l2(a, b): return ((a.double()-b.double()).norm()/b.double().norm()).item()
g = torch.Generator().manual_seed(0)
# 1) single ResBlock conv, K = 9216
x = torch.randn(1, 1024, 4, 8, generator=g); w = torch.randn(1024, 1024, 3, 3, generator=g) * 0.02
y64 = F.conv2d(F.pad(x.double(), (1,)*4, mode='reflect'), w.double())
y32 = F.conv2d(F.pad(x, (1,)*4, mode='reflect'), w)
L = HipConv2d(1024, 1024, 3, 1, 1, PAD_REFLECT, apply_bias=False, dtype=F32, device=DEV)
with torch.no_grad(): L.weight.copy_(w)
yh, _ = L.fwd(to_act(x, F32))
print('conv K=9216      HIP %.2e  torch32 %.2e' % (l2(to_nchw(yh), y64), l2(y32, y64)))
# 2) instance norm over 8 samples, large mean
x = torch.randn(1, 1024, 2, 4, generator=g) * 0.3 + 0.1
n64 = F.instance_norm(x.double(), eps=1e-5); n32 = F.instance_norm(x, eps=1e-5)
nh, _ = InstNormAct(ACT_NONE).fwd(to_act(x, F32))
print('inorm 8 samples  HIP %.2e  torch32 %.2e' % (l2(to_nchw(nh), n64), l2(n32, n64)))
x = torch.randn(1, 64, 32, 64, generator=g) * 0.3 + 0.1
n64 = F.instance_norm(x.double(), eps=1e-5); n32 = F.instance_norm(x, eps=1e-5)
nh, _ = InstNormAct(ACT_NONE).fwd(to_act(x, F32))
print('inorm 2048 samp  HIP %.2e  torch32 %.2e' % (l2(to_nchw(nh), n64), l2(n32, n64)))
# 3) whole generator, ngf=64 at 32x64
cfg = dict(netG='global', ngf=64, n_downsample_global=4, n_blocks_global=9, n_local_enhancers=1, n_blocks_local=3)
torch.manual_seed(1234)
sd = nets.init_generator(cfg, 39, 3)
net = networks.define_G(39, 3, 64, 'global', 4, 9, gpu_ids=[0]); net.load_state_dict(sd)
x = torch.rand(1, 39, 32, 64, generator=g) - 0.5
with torch.no_grad():
  y64 = nets.generator({k: v.double() for k, v in sd.items()}, x.double(), cfg)
  y32 = nets.generator(sd, x, cfg)
yh, _ = net.fwd(to_act(x, F32))
print('G ngf64 32x64    HIP %.2e  torch32 %.2e' % (l2(to_nchw(yh), y64), l2(y32, y64)))
