"""A/B of the InstanceNorm forms (developer build): single-kernel (mode 1) against moment -> finalize -> apply (mode 27) on the
tensors of the bench step.  Usage: python scripts/bench_norm.py [1,27]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch, jpdse_hip
from jpdse_hip import BF16, ACT_RELU, ACT_NONE, ops
from jpdse_hip.ops import Act
dev = torch.device('cuda', 0)
modes = [int(m) for m in (sys.argv[1] if len(sys.argv) > 1 else '1,27').split(',')]
SHAPES = [('resblock 1024@32x64', 4, 32, 64, 1024), ('down3 512@64x128', 4, 64, 128, 512),
          ('D s2 128@65x129 x8', 8, 65, 129, 128), ('D s2 512@34x66 x8', 8, 34, 66, 512),
          ('down2 256@128x256', 4, 128, 256, 256)]
def timeit(fn, iters=50):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters * 1e3
for name, N, H, W, C in SHAPES:
  x = Act((torch.randn(N, H, W, C, device=dev) * 1.5 + 0.5).bfloat16(), C)
  dy = Act(torch.randn(N, H, W, C, device=dev).bfloat16(), C)
  mb = N * H * W * C * 2 / 1e6
  for rep in range(2):
    for m in modes:
      jpdse_hip.set_dev_mode(m)
      y, stats = ops.inorm_fwd(x, ACT_RELU)
      tf = timeit(lambda: ops.inorm_fwd(x, ACT_RELU))
      tb = timeit(lambda: ops.inorm_bwd(x, stats, dy, ACT_RELU))
      print('%-22s %6.1f MB rep %d mode %2d: fwd %6.1f us   bwd %6.1f us' % (name, mb, rep, m, tf, tb), flush=True)
