"""InstanceNorm+act micro-benchmark: per-kernel-phase HBM rate at the bench workload's shapes."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch
from jpdse_hip import ops, BF16, ACT_RELU
from jpdse_hip.ops import Act
dev = torch.device('cuda', 0)
def timeit(fn, iters=20):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters
for (N, H, W, C) in [(4, 512, 1024, 64), (4, 256, 512, 128), (4, 128, 256, 256), (4, 64, 128, 512), (4, 32, 64, 1024), (8, 129, 257, 128)]:
  x = Act(torch.randn(N, H, W, C, device=dev).bfloat16(), C)
  dy = Act(torch.randn(N, H, W, C, device=dev).bfloat16(), C)
  y, stats = ops.inorm_fwd(x, ACT_RELU)
  tf = timeit(lambda: ops.inorm_fwd(x, ACT_RELU))
  tb = timeit(lambda: ops.inorm_bwd(x, stats, dy, ACT_RELU))
  nb = x.t.numel() * 2
  print('%-22s fwd %.3f ms (%.2f TB/s of 3 passes)   bwd %.3f ms (%.2f TB/s of 5 passes)' %
        ((N, H, W, C), tf, 3 * nb / tf / 1e9, tb, 5 * nb / tb / 1e9))
