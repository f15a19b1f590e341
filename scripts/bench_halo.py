"""A/B of halo-kernel variants (developer build): fwd + dgrad of the ResnetBlock conv and two VGG layers, timing and bit-comparison
against the first mode.  Usage: python scripts/bench_halo.py 1,25"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch, jpdse_hip
from jpdse_hip import BF16, PAD_REFLECT, PAD_ZERO, ACT_RELU, ACT_NONE
from jpdse_hip.layers import HipConv2d
from jpdse_hip.ops import Act
dev = torch.device('cuda', 0)
modes = [int(m) for m in sys.argv[1].split(',')]
LAYERS = [('resblock 1024 @32x64', 4, 32, 64, 1024, 1024, PAD_REFLECT, ACT_NONE),
          ('vgg conv4 512 @64x128', 4, 64, 128, 512, 512, PAD_ZERO, ACT_RELU),
          ('vgg conv1_2 64 @512x1024', 4, 512, 1024, 64, 64, PAD_ZERO, ACT_RELU),
          ('vgg conv2_1 64->128 @256x512', 4, 256, 512, 64, 128, PAD_ZERO, ACT_RELU)]
def timeit(fn, iters=20):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters
for name, N, H, W, C, K, mode, act in LAYERS:
  torch.manual_seed(0)
  layer = HipConv2d(C, K, 3, 1, 1, mode, act=act, apply_bias=(act != ACT_NONE), dtype=BF16, device=dev)
  x = Act(torch.randn(N, H, W, C, device=dev).bfloat16(), C)
  gfl = 2.0 * N * H * W * C * K * 9 / 1e9
  outs = {}
  for rep in range(3):
    for m in modes:
      jpdse_hip.set_dev_mode(m)
      y, ctx = layer.fwd(x)
      dy = Act(torch.randn(y.t.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1)).bfloat16(), K)
      tf = timeit(lambda: layer.fwd(x))
      td = timeit(lambda: layer.bwd(ctx, dy, True, False))
      dx = layer.bwd(ctx, dy, True, False)
      torch.cuda.synchronize()
      outs[m] = (y.t.clone(), dx.t.clone())
      print('%-30s rep %d mode %2d: fwd %.4f ms %5.0f TF   dgrad %.4f ms %5.0f TF' % (name, rep, m, tf, gfl / tf, td, gfl / td), flush=True)
  for m in modes[1:]:
    print('   mode %d vs %d: fwd identical %s, dgrad identical %s' % (m, modes[0], torch.equal(outs[m][0], outs[modes[0]][0]),
                                                                    torch.equal(outs[m][1], outs[modes[0]][1])))
