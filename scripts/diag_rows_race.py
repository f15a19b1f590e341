"""Race detector for the row-streaming kernels (counted vmcnt / lgkmcnt pipelines, LDS rings): repeat fwd + backward of the layers
they serve AT THE BENCH SIZE and compare bitwise with the first result, with allocator / cache disturbance between calls."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch
from jpdse_hip import BF16, PAD_ZERO, PAD_REFLECT, ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH
from jpdse_hip.layers import HipConv2d
from jpdse_hip.ops import Act
dev = torch.device('cuda', 0)
cases = [('down 64->128 s2', 4, 512, 1024, 64, 128, 3, 2, 1, PAD_ZERO, ACT_NONE, False),
         ('vgg conv1_2', 4, 512, 1024, 64, 64, 3, 1, 1, PAD_ZERO, ACT_RELU, False),
         ('vgg conv2_1', 4, 256, 512, 64, 128, 3, 1, 1, PAD_ZERO, ACT_RELU, False),
         ('convT 128->64', 4, 256, 512, 128, 64, 3, 2, 1, PAD_ZERO, ACT_NONE, True),
         ('head 7x7', 4, 512, 1024, 64, 3, 7, 1, 3, PAD_REFLECT, ACT_TANH, False),
         ('vgg conv1_1', 4, 512, 1024, 3, 64, 3, 1, 1, PAD_ZERO, ACT_RELU, False),
         ('D layer0', 8, 512, 1024, 39, 64, 4, 2, 2, PAD_ZERO, ACT_LRELU, False)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for (name, N, H, W, C, K, k, st, pad, mode, act, tr) in cases:
  torch.manual_seed(5)
  layer = HipConv2d(C, K, k, st, pad, mode, act=act, apply_bias=(act != ACT_NONE and not tr), transposed=tr, dtype=BF16, device=dev)
  Cs = (C + 7) // 8 * 8
  xt = torch.randn(N, H, W, Cs, device=dev).bfloat16()
  if Cs != C: xt[..., C:] = 0
  x = Act(xt, C)
  y0, ctx = layer.fwd(x)
  dy = Act(torch.randn(y0.t.shape, device=dev).bfloat16(), y0.C)
  if y0.t.shape[-1] != y0.C: dy.t[..., y0.C:] = 0
  dx0 = layer.bwd(ctx, dy, True, True).t.clone()
  dw0 = layer.weight.grad.clone()
  ds0 = layer.bwd_input_slice(ctx, dy, 36, 39).t.clone() if name == 'D layer0' else None
  y0 = y0.t.clone()
  bad = [0, 0, 0, 0]
  for it in range(reps):
    junk = torch.full((64 << 20,), float('nan'), device=dev)      # 256 MB: flushes the Infinity Cache too
    y, ctx = layer.fwd(x)
    dx = layer.bwd(ctx, dy, True, True)
    bad[0] += int(not torch.equal(y.t, y0))
    bad[1] += int(not torch.equal(dx.t, dx0))
    bad[2] += int(not torch.equal(layer.weight.grad, dw0))
    if ds0 is not None:
      bad[3] += int(not torch.equal(layer.bwd_input_slice(ctx, dy, 36, 39).t, ds0))
    del junk
  print('%-16s fwd mismatches %d, dgrad %d, wgrad %d, slice %d of %d' % (name, bad[0], bad[1], bad[2], bad[3], reps), flush=True)
