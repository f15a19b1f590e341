"""Host-side (Python) profile of the train step: where the CPU time per step goes, and how far the launch thread runs ahead
of the GPU.  Usage: python scripts/host_profile.py [--steps 5]"""
import sys, os, argparse, cProfile, pstats, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch
ap = argparse.ArgumentParser()
ap.add_argument('--steps', type=int, default=5)
ap.add_argument('--batch', type=int, default=4)
args = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
import jpdse_hip
from ctu.trainers import get_trainer
from ctu.utils.synthetic import synthetic_batch, default_opt
opt = default_opt(gpu_ids=[0], print_losses=False, compute_dtype='bf16', use_compressed=True, netG='global', ngf=64,
                  batch_size=args.batch)
torch.manual_seed(1234)
trainer = get_trainer(opt)(opt, 'train')
xd = synthetic_batch(args.batch, 512, 1024, seed=1234)
xd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in xd.items()}
for _ in range(3):
  trainer.step(xd)
torch.cuda.synchronize()
# (1) pure host time: how long the launch thread needs for one step when it never waits for the GPU mid-step
t0 = time.perf_counter()
for _ in range(args.steps):
  trainer.step(xd)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / args.steps
pr = cProfile.Profile()
pr.enable()
for _ in range(args.steps):
  trainer.step(xd)
pr.disable()
torch.cuda.synchronize()
print('wall per step %.2f ms' % (wall * 1e3))
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
