"""Repeat fwd / dgrad of a few conv cases and compare bitwise with the first result (race detector)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch
from jpdse_hip import F32, BF16, PAD_ZERO, PAD_REFLECT, ACT_NONE, ACT_RELU
from jpdse_hip.layers import HipConv2d
from jpdse_hip.ops import Act
dev = torch.device('cuda', 0)
cases = [(2, 5, 128, 256, 256, 3, 1, 1, PAD_ZERO, ACT_RELU), (1, 6, 64, 128, 256, 3, 1, 1, PAD_REFLECT, ACT_NONE),
         (2, 8, 64, 64, 128, 3, 1, 1, PAD_REFLECT, ACT_NONE), (2, 9, 13, 128, 256, 4, 2, 2, PAD_ZERO, ACT_NONE)]
for dtype, tdt in ((F32, torch.float32), (BF16, torch.bfloat16)):
  for (N, H, W, C, K, k, st, pad, mode, act) in cases:
    torch.manual_seed(5)
    layer = HipConv2d(C, K, k, st, pad, mode, act=act, apply_bias=True, dtype=dtype, device=dev)
    x = Act(torch.randn(N, H, W, C, device=dev).to(tdt), C)
    y0, ctx = layer.fwd(x)
    dy = Act(torch.randn(y0.t.shape, device=dev).to(tdt), y0.C)
    dx0 = layer.bwd(ctx, dy, True, True).t.clone()
    y0 = y0.t.clone()
    bad = [0, 0]
    for it in range(150):
      # disturb the shared workspace / allocator between calls like a test sequence would
      junk = torch.full((1 << 20,), float('nan'), device=dev)
      y, ctx = layer.fwd(x)
      dx = layer.bwd(ctx, dy, True, True)
      bad[0] += int(not torch.equal(y.t, y0))
      bad[1] += int(not torch.equal(dx.t, dx0))
      del junk
    print('dtype %d case %s: fwd mismatches %d, dgrad mismatches %d of 150' % (dtype, (N, H, W, C, K, k, st), bad[0], bad[1]))
