"""fast (split-K) vs generic bf16 conv on identical inputs: outputs may differ by output rounding only."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch
import jpdse_hip
from jpdse_hip import lib, BF16, PAD_ZERO, PAD_REFLECT, ACT_NONE
from jpdse_hip.layers import HipConv2d
from jpdse_hip.ops import Act
dev = torch.device('cuda', 0)
for (N, H, W, C, K, k, st, pad, mode) in [(2, 8, 16, 1024, 1024, 3, 1, 1, PAD_REFLECT), (1, 15, 32, 256, 256, 3, 1, 1, PAD_REFLECT),
                                          (2, 16, 32, 256, 512, 3, 2, 1, PAD_ZERO), (4, 16, 32, 1024, 1024, 3, 1, 1, PAD_REFLECT)]:
  torch.manual_seed(3)
  layer = HipConv2d(C, K, k, st, pad, mode, apply_bias=False, dtype=BF16, device=dev)
  x = Act(torch.randn(N, H, W, C, device=dev).bfloat16(), C)
  out = {}
  for m in (1, 6, 0):
    jpdse_hip.set_dev_mode(m)
    y, ctx = layer.fwd(x)
    dy = Act((torch.arange(y.t.numel(), device=dev).reshape(y.t.shape) % 7 - 3).bfloat16(), y.C)
    dx = layer.bwd(ctx, dy, True, False)
    out[m] = (y.t.float().clone(), dx.t.float().clone())
  jpdse_hip.set_dev_mode(1)
  for m in (1, 6):
    for i, nm in enumerate(('fwd', 'dgrad')):
      a, b = out[m][i], out[0][i]
      d = (a - b).abs()
      print('%s mode%d vs generic %s: max|diff| %.4g  mean|diff| %.4g  max|ref| %.4g  rel-L2 %.3e  frac differing %.4f' %
            ((N, H, W, C, K, st), m, nm, d.max().item(), d.mean().item(), b.abs().max().item(), ((a - b).norm() / b.norm()).item(), (d > 0).float().mean().item()))
