"""Checksums of the InstanceNorm forward / backward outputs on fixed random inputs, for comparing two library builds bit for bit:
   JPDSE_HIP_LIB=<other .so> python scripts/diag_norm_hash.py"""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch, jpdse_hip
from jpdse_hip import ops, BF16, F32, ACT_NONE, ACT_RELU, ACT_LRELU
from jpdse_hip.ops import Act
dev = torch.device('cuda', 0)
def h(t): return hashlib.md5(t.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes()).hexdigest()[:10]
for dt, tdt in ((BF16, torch.bfloat16), (F32, torch.float32)):
  for (N, H, W, C) in ((2, 32, 64, 256), (2, 128, 256, 64), (1, 512, 1024, 64)):
    g = torch.Generator(device=dev).manual_seed(5)
    x = Act((torch.randn(N, H, W, C, device=dev, generator=g) * 1.3 + 0.2).to(tdt), C)
    r = Act(torch.randn(N, H, W, C, device=dev, generator=g).to(tdt), C)
    dy = Act(torch.randn(N, H, W, C, device=dev, generator=g).to(tdt), C)
    for act in (ACT_NONE, ACT_RELU, ACT_LRELU):
      for res in (None, r):
        if res is not None and act != ACT_NONE: continue
        y, st = ops.inorm_fwd(x, act, residual=res)
        dx = ops.inorm_bwd(x, st, dy, act)
        torch.cuda.synchronize()
        print('dt %d %s act %d res %d: y %s stats %s dx %s' % (dt, (N, H, W, C), act, res is not None, h(y.t), h(st), h(dx.t)))
