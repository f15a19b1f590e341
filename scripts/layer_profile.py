"""Per-conv-call profile of one real train step at the bench workload: wraps ops.conv_fwd /
conv_dgrad / conv_wgrad with event pairs (includes their pad / fold / expand helper kernels) and
prints every distinct (op, shape) with its FLOPs, time and TFLOP/s, sorted by time.
Usage: python scripts/layer_profile.py [--batch 4] [--steps 3] [--fast MODE]"""
import sys, os, argparse, collections, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=4)
ap.add_argument('--steps', type=int, default=3)
ap.add_argument('--width', type=int, default=1024)
ap.add_argument('--height', type=int, default=512)
ap.add_argument('--netG', default='global')
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--fast', type=int, default=1)
ap.add_argument('--no-vgg', action='store_true', help='BASELINE config 2: no VGG loss, VGG not run')
args = ap.parse_args()

torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
import jpdse_hip
from jpdse_hip import lib, ops
from ctu.trainers import get_trainer
from ctu.utils.synthetic import synthetic_batch, default_opt

jpdse_hip.set_dev_mode(args.fast)
opt = default_opt(gpu_ids=[0], print_losses=False, compute_dtype=args.dtype, use_compressed=True,
                  netG=args.netG, ngf=64 if args.netG == 'global' else 32, batch_size=args.batch,
                  **(dict(no_vgg_loss=True, skip_unused_losses=True) if args.no_vgg else {}))
torch.manual_seed(1234)
trainer = get_trainer(opt)(opt, 'train')
xd = synthetic_batch(args.batch, args.height, args.width, seed=1234)
xd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in xd.items()}
for _ in range(2):
  trainer.step(xd)
torch.cuda.synchronize()

records = []          # (key, e0, e1)
_orig = dict(fwd=ops.conv_fwd, dgrad=ops.conv_dgrad, wgrad=ops.conv_wgrad, fwd_pool=ops.conv_fwd_pool)


def _key(kind, d):
  return (kind, d.N, d.H, d.W, d.C, d.K, d.R, d.stride, d.pad_mode, d.pad)


def _wrap(kind):
  f = _orig[kind]
  def g(d, *a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    c0 = time.perf_counter()
    r = f(d, *a, **k)
    c1 = time.perf_counter()
    e1.record()
    records.append((_key('fwd' if kind == 'fwd_pool' else kind, d), e0, e1, c1 - c0))     # conv + pooled epilogue counts as the forward
    return r
  return g


ops.conv_fwd, ops.conv_dgrad, ops.conv_wgrad = _wrap('fwd'), _wrap('dgrad'), _wrap('wgrad')
ops.conv_fwd_pool = _wrap('fwd_pool')
# convs whose epilogue also writes the InstanceNorm moments (ConvTranspose forward = the data gradient of the underlying conv)
_fm = ops.conv_fwd_moments
def _fwd_moments(d, x, pack, slots, transposed=False):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  c0 = time.perf_counter()
  r = _fm(d, x, pack, slots, transposed)
  c1 = time.perf_counter()
  e1.record()
  records.append((_key('dgrad' if transposed else 'fwd', d), e0, e1, c1 - c0))
  return r
ops.conv_fwd_moments = _fwd_moments
# data gradients whose epilogue also writes the sums of the consuming InstanceNorm's backward (round 4)
_dn = ops.conv_dgrad_nsums
def _dgrad_nsums(d, *a, **k):
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  c0 = time.perf_counter()
  r = _dn(d, *a, **k)
  c1 = time.perf_counter()
  e1.record()
  records.append((_key('dgrad', d), e0, e1, c1 - c0))
  return r
ops.conv_dgrad_nsums = _dgrad_nsums
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(args.steps):
  trainer.step(xd)
t1.record()
torch.cuda.synchronize()
total = t0.elapsed_time(t1) / args.steps

agg = collections.OrderedDict()
for key, e0, e1, cpu in records:
  a = agg.setdefault(key, [0, 0.0, 0.0])
  a[0] += 1; a[1] += e0.elapsed_time(e1); a[2] += cpu * 1e3


def flops(key):
  kind, N, H, W, C, K, R, st, pm, pad = key
  OH = (H + 2 * pad - R) // st + 1; OW = (W + 2 * pad - R) // st + 1
  return 2.0 * N * OH * OW * C * K * R * R


rows = []
for key, (n, ms, cpu) in agg.items():
  n_step = n / args.steps; ms_step = ms / args.steps
  rows.append((ms_step, key, n_step, flops(key) * n_step, cpu / args.steps))
rows.sort(reverse=True)
conv_ms = sum(r[0] for r in rows); conv_fl = sum(r[3] for r in rows)
print('step %.2f ms (with event overhead); conv ops %.2f ms, %.1f GFLOP -> %.0f TFLOP/s avg'
      % (total, conv_ms, conv_fl / 1e9, conv_fl / conv_ms / 1e9))
print('%-6s %3s %5s %5s %5s %5s %2s %2s %3s | %4s %8s %8s %7s %6s' %
      ('op', 'N', 'H', 'W', 'C', 'K', 'R', 's', 'pad', 'n', 'ms/step', 'GFLOP', 'TFLOP/s', 'lost'))
for ms, key, n, fl, cpu in rows:
  kind, N, H, W, C, K, R, st, pm, pad = key
  tf = fl / ms / 1e9
  lost = ms - fl / 1e12      # ms above a 1000 TFLOP/s pace
  print('%-6s %3d %5d %5d %5d %5d %2d %2d %3s | %4.0f %8.3f %8.1f %7.0f %6.2f  cpu %.3f' %
        (kind, N, H, W, C, K, R, st, 'rfl' if pm == 1 else 'zer', n, ms, fl / 1e9, tf, lost, cpu))
