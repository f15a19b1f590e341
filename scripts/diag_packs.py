"""Which conv layers still take the lazy per-layer panel pack inside a train step (and which kernels it launches)."""
import sys, os, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch
sys.argv = [sys.argv[0], '--no-cpu-baseline']
import bench
args = bench.parse()
opt = bench.make_opt(args, 0)
from ctu.trainers import get_trainer
from ctu.utils.synthetic import synthetic_batch
from jpdse_hip import ops, layers
import contextlib
with contextlib.redirect_stdout(sys.stderr):
  trainer = get_trainer(opt)(opt, 'train')
xd = synthetic_batch(args.batch, args.height, args.width, seed=1)
xd = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in xd.items()}
for _ in range(3):
  trainer.step(xd)
names = {}
for tag, net in (('G', trainer.model.netG), ('D', trainer.model.netD)):
  for n, m in net.named_modules():
    if isinstance(m, layers.HipConv2d):
      names[id(m)] = tag + '.' + n
count = collections.Counter()
orig_into, orig_pack = ops.conv_pack_into, ops.conv_pack
import inspect
def wrap(f, label):
  def g(*a, **k):
    fr = inspect.currentframe().f_back
    while fr is not None and not isinstance(fr.f_locals.get('self'), layers.HipConv2d):
      fr = fr.f_back
    who = names.get(id(fr.f_locals['self']), '?') if fr is not None else '?'
    count[(label, who)] += 1
    return f(*a, **k)
  return g
ops.conv_pack_into = wrap(orig_into, 'pack_into')
ops.conv_pack = wrap(orig_pack, 'pack')
trainer.step(xd)
torch.cuda.synchronize()
for k, v in sorted(count.items()):
  print(v, k)
