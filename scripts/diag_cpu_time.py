"""CPU enqueue time of one train step (before the loss readback) vs GPU time."""
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
for _ in range(3):
  tr.step(xd)
torch.cuda.synchronize()
real_cpu = torch.Tensor.cpu
marks = []
def cpu_hook(self, *a, **k):
  marks.append(time.perf_counter())
  return real_cpu(self, *a, **k)
torch.Tensor.cpu = cpu_hook
for i in range(5):
  torch.cuda.synchronize()
  t0 = time.perf_counter(); marks.clear()
  tr.step(xd)
  t2 = time.perf_counter()
  print('step %d: CPU enqueue until the loss readback %.2f ms, whole step %.2f ms' % (i, 1e3 * (marks[0] - t0), 1e3 * (t2 - t0)))
