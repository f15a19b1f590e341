import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/jpd-se_amd')
import torch, jpdse_hip
from jpdse_hip import BF16, PAD_REFLECT
from jpdse_hip.layers import HipConv2d
from jpdse_hip.ops import Act
dev = torch.device('cuda', 0)
torch.manual_seed(0)
layer = HipConv2d(1024, 1024, 3, 1, 1, PAD_REFLECT, apply_bias=False, dtype=BF16, device=dev)
x = Act(torch.randn(4, 32, 64, 1024, device=dev).bfloat16(), 1024)
y, ctx = layer.fwd(x)
dy = Act(torch.randn_like(y.t.float()).bfloat16(), 1024)
res = {}
def timeit(fn, iters=20):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters
modes = [int(m) for m in sys.argv[1].split(',')]
grads = {}
for rep in range(3):
  for m in modes:
    jpdse_hip.set_dev_mode(m)
    t = timeit(lambda: layer.bwd(ctx, dy, False, True))
    grads[m] = layer.weight.grad.detach().clone()
    print('rep %d mode %d: wgrad %.4f ms  %.0f TFLOP/s' % (rep, m, t, 154.6 / t), flush=True)
base = grads[modes[0]]
for m in modes[1:]:
  print('mode %d vs %d: max abs diff %.3e (bit-identical: %s)' % (m, modes[0], (grads[m] - base).abs().max().item(), torch.equal(grads[m], base)))
