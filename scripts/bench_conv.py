"""Per-layer micro-benchmark of the conv kernels (fwd / dgrad / wgrad) at the bench workload's shapes.
Usage: python scripts/bench_conv.py [--batch 4] [--fast 0|1] [--filter name]"""
import sys, os, argparse, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch
import jpdse_hip
from jpdse_hip import lib, BF16, F32, PAD_ZERO, PAD_REFLECT, ACT_NONE
from jpdse_hip.layers import HipConv2d
from jpdse_hip.ops import Act

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--fast', type=str, default='1', help='comma list of fast-path modes to A/B in one process')
ap.add_argument('--filter', default='')
ap.add_argument('--iters', type=int, default=10)
ap.add_argument('--fwd-only', action='store_true', help='time the forward launch only (long steady runs for scripts/clock_probe.sh)')
args = ap.parse_args()
dev = torch.device('cuda', 0)
modes = [int(x) for x in args.fast.split(',')]
B = args.batch
# name, H, W, C, K, k, stride, pad, mode, transposed
LAYERS = [
  ('G first 7x7 39->64 @512x1024', 512, 1024, 39, 64, 7, 1, 3, PAD_REFLECT, False),
  ('G down 64->128 s2',            512, 1024, 64, 128, 3, 2, 1, PAD_ZERO, False),
  ('G down 128->256 s2',           256, 512, 128, 256, 3, 2, 1, PAD_ZERO, False),
  ('G down 256->512 s2',           128, 256, 256, 512, 3, 2, 1, PAD_ZERO, False),
  ('G down 512->1024 s2',          64, 128, 512, 1024, 3, 2, 1, PAD_ZERO, False),
  ('ResBlock 1024 3x3 @32x64',     32, 64, 1024, 1024, 3, 1, 1, PAD_REFLECT, False),
  ('G up convT 1024->512',         32, 64, 1024, 512, 3, 2, 1, PAD_ZERO, True),
  ('G up convT 512->256',          64, 128, 512, 256, 3, 2, 1, PAD_ZERO, True),
  ('G up convT 256->128',          128, 256, 256, 128, 3, 2, 1, PAD_ZERO, True),
  ('G up convT 128->64',           256, 512, 128, 64, 3, 2, 1, PAD_ZERO, True),
  ('G last 7x7 64->3',             512, 1024, 64, 3, 7, 1, 3, PAD_REFLECT, False),
  ('VGG conv1_1 3->64',            512, 1024, 3, 64, 3, 1, 1, PAD_ZERO, False),
  ('VGG conv1_2 64->64',           512, 1024, 64, 64, 3, 1, 1, PAD_ZERO, False),
  ('VGG conv2_1 64->128',          256, 512, 64, 128, 3, 1, 1, PAD_ZERO, False),
  ('VGG conv2_2 128->128',         256, 512, 128, 128, 3, 1, 1, PAD_ZERO, False),
  ('VGG conv3_x 256->256',         128, 256, 256, 256, 3, 1, 1, PAD_ZERO, False),
  ('VGG conv4_x 512->512',         64, 128, 512, 512, 3, 1, 1, PAD_ZERO, False),
  ('L ResBlock 64 3x3 @256x512',    256, 512, 64, 64, 3, 1, 1, PAD_REFLECT, False),
  ('D layer4 512->1 4x4 s1 @66x130 N8', 66, 130, 512, 1, 4, 1, 2, PAD_ZERO, False),
  ('D layer0 39->64 4x4 s2',       512, 1024, 39, 64, 4, 2, 2, PAD_ZERO, False),
  ('D layer1 64->128 4x4 s2',      257, 513, 64, 128, 4, 2, 2, PAD_ZERO, False),
  ('D layer2 128->256 4x4 s2',     129, 257, 128, 256, 4, 2, 2, PAD_ZERO, False),
  ('D layer3 256->512 4x4 s1',     65, 129, 256, 512, 4, 1, 2, PAD_ZERO, False),
  ('D1 layer1 64->128 4x4 s2',     129, 257, 64, 128, 4, 2, 2, PAD_ZERO, False),
  ('D1 layer2 128->256 4x4 s2',    65, 129, 128, 256, 4, 2, 2, PAD_ZERO, False),
  ('D1 layer3 256->512 4x4 s1',    33, 65, 256, 512, 4, 1, 2, PAD_ZERO, False),
]

def timeit(fn, iters):
  fn(); torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(iters): fn()
  e1.record(); torch.cuda.synchronize()
  return e0.elapsed_time(e1) / iters

def _bench_one(name, layer, H, W, C, K, k, tr):
  B = 2 * args.batch if name.startswith('D') else args.batch     # the discriminator sees [fake ; real]
  x = Act(torch.randn(B, H, W, (C + 7) // 8 * 8, device=dev).bfloat16(), C)
  y, ctx = layer.fwd(x)
  dy = Act(torch.randn_like(y.t.float()).bfloat16(), y.C)
  macs = B * H * W * C * K * 9 if tr else B * y.H * y.W * K * C * k * k
  fl = 2.0 * macs
  ts = []
  for rep in range(2):      # interleaved repeats: take the best of two
    t_f = timeit(lambda: layer.fwd(x), args.iters)
    t_d = t_f if args.fwd_only else timeit(lambda: layer.bwd(ctx, dy, True, False), args.iters)
    t_w = t_f if args.fwd_only else timeit(lambda: layer.bwd(ctx, dy, False, True), args.iters)
    ts.append((t_f, t_d, t_w))
  t_f, t_d, t_w = (min(t[i] for t in ts) for i in range(3))
  print('%-34s %5.0f %5.2f %5.0f %5.2f %5.0f %5.2f' % (name, fl / t_f / 1e9, t_f, fl / t_d / 1e9, t_d, fl / t_w / 1e9, t_w))

print('%-34s %10s %10s %10s   (TFLOP/s, ms)' % ('layer', 'fwd', 'dgrad', 'wgrad'))
for (name, H, W, C, K, k, st, pad, mode, tr) in LAYERS:
  if args.filter and not any(f in name for f in args.filter.split(',')): continue
  for fm in modes:
   jpdse_hip.set_dev_mode(fm)
   layer = HipConv2d(C, K, k, st, pad, mode, apply_bias=False, transposed=tr, dtype=BF16, device=dev)
   name_m = '%s [m%d]' % (name[:28], fm)
   _bench_one(name_m, layer, H, W, C, K, k, tr)
