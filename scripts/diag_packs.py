import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch, collections
from jpdse_hip import ops
from ctu.trainers import get_trainer
from ctu.utils import synthetic as omodel
opt = omodel.default_opt(gpu_ids=[0], print_losses=False, compute_dtype='bf16', ngf=16)
tr = get_trainer(opt)(opt, 'train')
xd = omodel.synthetic_batch(1, 64, 128, seed=21)
cnt = collections.Counter()
orig = ops.conv_pack_into
def wrapped(d, w, fwd, dgr):
  cnt['fwd+dgrad' if fwd is not None else 'dgrad only'] += 1
  return orig(d, w, fwd, dgr)
ops.conv_pack_into = wrapped
import jpdse_hip.layers as L
for i in range(3):
  tr.step(xd)
  print('step', i, dict(cnt)); cnt.clear()
