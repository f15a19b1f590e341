"""Hunt for the intermittent fp32 dgrad mismatch: alternate a bf16 case and an fp32 case with fresh layers."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from jpdse_hip import F32, BF16, PAD_ZERO, PAD_REFLECT, ACT_NONE, ACT_RELU
from jpdse_hip.layers import HipConv2d
from jpdse_hip.ops import Act
from hip_util import to_act, to_nchw, quantize_like
dev = torch.device('cuda', 0)

def run(case, dtype, seed):
  N, H, W, C, K, k, st, pad, mode, act = case
  g = torch.Generator().manual_seed(seed)
  x = quantize_like(torch.randn(N, C, H, W, generator=g), dtype)
  w = torch.randn(K, C, k, k, generator=g) * (1.0 / (C * k * k) ** 0.5)
  b = torch.randn(K, generator=g) * 0.1
  layer = HipConv2d(C, K, k, st, pad, mode, act=act, apply_bias=True, dtype=dtype, device=dev)
  with torch.no_grad():
    layer.weight.copy_(w); layer.bias.copy_(b)
  y, ctx = layer.fwd(to_act(x, dtype))
  gy = quantize_like(torch.randn(to_nchw(y).shape, generator=g), dtype)
  dx = layer.bwd(ctx, to_act(gy, dtype), need_dx=True, need_dw=True)
  torch.cuda.synchronize()
  return to_nchw(y).clone(), to_nchw(dx).clone(), layer.weight.grad.detach().cpu().clone()

A = (2, 8, 64, 128, 64, 3, 1, 1, PAD_ZERO, ACT_RELU)      # halo_vgg_64
B = (1, 12, 128, 64, 256, 3, 1, 1, PAD_ZERO, ACT_RELU)    # halo_vgg_256
C2 = (1, 6, 64, 128, 256, 3, 1, 1, PAD_REFLECT, ACT_NONE) # wgrad_row_refl
D = (2, 5, 128, 256, 256, 3, 1, 1, PAD_ZERO, ACT_RELU)    # wgrad_row_zero
ref = {}
for it in range(60):
  for (pre, pre_dt, cur, name) in ((A, BF16, B, 'B'), (C2, BF16, D, 'D')):
    run(pre, pre_dt, 11)
    y, dx, dw = run(cur, F32, 12)
    if name not in ref:
      ref[name] = (y, dx, dw)
      continue
    for nm, a, b in (('fwd', y, ref[name][0]), ('dgrad', dx, ref[name][1]), ('wgrad', dw, ref[name][2])):
      if not torch.equal(a, b):
        d = (a - b).abs()
        idx = torch.nonzero(d > 0)
        print('iter %d case %s %s: %d of %d elements differ, max %.3e; first %s; distinct n,c,h,w ranges: %s' %
              (it, name, nm, idx.shape[0], a.numel(), d.max().item(), idx[0].tolist(),
               [(int(idx[:, j].min()), int(idx[:, j].max())) for j in range(idx.shape[1])]))
print('done')
