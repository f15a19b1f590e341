"""One-GPU rehearsal of data parallelism's CU contention: the bench step runs while a persistent
"occupier" kernel on a second stream holds k compute units (developer build: jpdse_debug_occupy_cus -- workgroups that own a
CU's whole LDS and sleep), standing in for the CUs a concurrent RCCL all-reduce keeps busy.  Per k: ms/step, the per-launch
time of the 256-workgroup kernels (gemm_halo / wgrad_nine on the ResnetBlock shape: one workgroup per CU, so a missing CU
means a second round) and the time of the step's sections (forward, G backward, D backward, optimizers).
Usage: python scripts/cu_contention.py [--ks 0,8,16,32] [--steps 10]   -> table on stdout (profiles/r03_cu_contention.txt)"""
import sys, os, argparse, ctypes, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
import torch

ap = argparse.ArgumentParser()
ap.add_argument('--ks', default='0,8,16,32')
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--batch', type=int, default=4)
args = ap.parse_args()
torch.cuda.set_device(0)
dev = torch.device('cuda', 0)
import jpdse_hip
from jpdse_hip import check
from ctu.trainers import get_trainer
from ctu.utils.synthetic import synthetic_batch, default_opt

jpdse_hip.set_dev_mode(1)          # developer build, shipped kernel selection
L = jpdse_hip.lib()
opt = default_opt(gpu_ids=[0], print_losses=False, compute_dtype='bf16', use_compressed=True, netG='global', ngf=64,
                  batch_size=args.batch)
torch.manual_seed(1234)
trainer = get_trainer(opt)(opt, 'train')
xd = synthetic_batch(args.batch, 512, 1024, seed=1234)
xd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in xd.items()}
for _ in range(3):
  trainer.step(xd)
torch.cuda.synchronize()

# section timers: events around the model's phases
m = trainer.model
sections = {}
def timed(name, fn):
  def g(*a, **k):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = fn(*a, **k)
    e1.record()
    sections.setdefault(name, []).append((e0, e1))
    return r
  return g
m._forward_losses = timed('forward + losses', m._forward_losses)
m.backward_G = timed('G backward', m.backward_G)
m.backward_D = timed('D backward', m.backward_D)
trainer.optimizer_G.step = timed('Adam G', trainer.optimizer_G.step)
trainer.optimizer_D.step = timed('Adam D', trainer.optimizer_D.step)

side = torch.cuda.Stream()
flag = torch.zeros(1, dtype=torch.int32).pin_memory()
rows = []
for k in [int(t) for t in args.ks.split(',')]:
  sections.clear()
  flag[0] = 0
  torch.cuda.synchronize()
  if k > 0:
    check(L.jpdse_debug_occupy_cus(k, ctypes.c_void_p(flag.data_ptr()), 15000, ctypes.c_void_p(side.cuda_stream)), 'occupy_cus')
    time.sleep(0.05)                 # the occupier's workgroups are resident before the step's kernels arrive
  check(L.jpdse_prof_select(1, 1024, 9216, 96 * args.steps), 'prof_select')
  t0 = time.perf_counter()
  for _ in range(args.steps):
    trainer.step(xd)
  torch.cuda.current_stream().synchronize()
  el = time.perf_counter() - t0
  flag[0] = 1                        # release the occupier
  torch.cuda.synchronize()
  out = {}
  for cls, name in ((1, 'ring'), (2, 'wgrad_nine')):
    ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    check(L.jpdse_prof_collect_class(cls, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)), 'collect')
    out[name] = (ms.value, n.value)
  ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
  check(L.jpdse_prof_collect(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)), 'collect')
  out['halo'] = (ms.value, n.value)
  check(L.jpdse_prof_select(0, 0, 0, 0), 'off')
  sec = {name: sum(a.elapsed_time(b) for a, b in ev) / args.steps for name, ev in sections.items()}
  rows.append((k, 1e3 * el / args.steps, out, sec))

print('bench step (1024x512, global ngf 64, batch %d, bf16) with k CUs held by the occupier; %d steps each, one process' % (args.batch, args.steps))
print('%4s %9s %8s | %12s %12s %10s | %s' % ('k', 'ms/step', 'vs k=0', 'halo us/launch', 'nine us/launch', 'ring us', 'sections (ms/step)'))
base = rows[0][1]
names = ['forward + losses', 'G backward', 'Adam G', 'D backward', 'Adam D']
for k, ms, out, sec in rows:
  us = lambda key: 1e3 * out[key][0] / max(out[key][1], 1)
  print('%4d %9.3f %7.1f%% | %12.1f %12.1f %10.1f | %s' % (k, ms, 100.0 * (ms / base - 1.0), us('halo'), us('wgrad_nine'), us('ring'),
        ', '.join('%s %.2f' % (nm, sec.get(nm, 0.0)) for nm in names)))
