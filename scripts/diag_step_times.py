"""Per-step wall time over a longer run (does the sustained rate differ from the first steps?)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch
from ctu.trainers import get_trainer
from ctu.utils.synthetic import synthetic_batch, default_opt
dev = torch.device('cuda', 0)
opt = default_opt(gpu_ids=[0], print_losses=False, compute_dtype='bf16', use_compressed=True, batch_size=4)
torch.manual_seed(1234)
tr = get_trainer(opt)(opt, 'train')
xd = synthetic_batch(4, 512, 1024, seed=1234)
xd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in xd.items()}
import gc
if len(sys.argv) > 1 and sys.argv[1] == "nogc":
  gc.collect(); gc.freeze(); gc.disable()
ts = []
for i in range(60):
  torch.cuda.synchronize(); t0 = time.perf_counter()
  tr.step(xd)
  torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print('step ms:', ' '.join('%.1f' % t for t in ts))
