#!/usr/bin/env python
"""Headline benchmark: training images/sec of one full JPD-SE `Pix2PixHDTrainer.step`
(G fwd/bwd + batched 2-scale PatchGAN + 2x VGG19 + both Adam updates) at 1024x512.

  python bench.py --gpus N --steps K --warmup W
N > 1 works both ways: under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or invoked directly, in which case this process
-- BEFORE it makes any GPU call -- starts that very launcher as a child, relays rank 0's JSON line and exits with the
child's code (no exec of a process that has touched the GPU).

Workload (BASELINE.json metric / configs[3] at N GPUs): script-default
GlobalGenerator ngf=64, 4 downsamples, 9 ResnetBlocks; 39 input channels; num_D=2; LSGAN +
D-feature-matching + VGG19 + L1; batch 4 per GPU (weak scaling), bf16 MFMA inputs with fp32
accumulation, fp32 master weights/Adam; synthetic Cityscapes-shaped inputs resident in HBM.
One JSON line on rank 0.  `roofline`: the ResnetBlock 3x3 implicit-GEMM kernel (N=1024,
K=9216), algorithmic FLOPs / hipEvent-measured kernel time inside the timed region, against the
2.5 PFLOP/s dense bf16 MFMA peak.  `cpu_baseline`: the torch-CPU oracle stepping the same
network on the host cores (bounded sample; a reported baseline, not the target).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, 'jpd-se_amd')):
  if _p not in sys.path:
    sys.path.insert(0, _p)

import torch
import torch.distributed as dist

# algorithmic GFLOP per image of the train step (2*MAC; G fwd+dgrad+wgrad,
# D(fake) & D(real) fwd+dgrad+wgrad, D(fake) for G fwd+dgrad, VGG(fake) fwd+dgrad, VGG(real) fwd)
F_ALG_GFLOP = {('global', 1024, 512): 4592.5, ('local', 1024, 512): 2803.2,
               ('global', 512, 256): 1154.3, ('global', 256, 128): 291.7,
               ('global', 2048, 1024): 18322.1, ('local', 2048, 1024): 11164.5}
MFMA_PEAK_TFLOPS = {'bf16': 2500.0, 'fp32': 157.3}     # MI355X_MICROARCH.md, dense
HBM_PEAK_GBS = 8000.0                                   # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; ~6.3 TB/s measured on a copy)


def parse():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=8)
  ap.add_argument('--warmup', type=int, default=2)
  ap.add_argument('--width', type=int, default=1024)
  ap.add_argument('--height', type=int, default=512)
  ap.add_argument('--batch', type=int, default=4, help='images per GPU')
  ap.add_argument('--netG', default='global', choices=['global', 'local'])
  ap.add_argument('--dtype', default='bf16', choices=['bf16', 'fp32'])
  ap.add_argument('--no-vgg', action='store_true',
                  help='BASELINE config 2: GlobalGenerator + 2-scale PatchGAN step without the VGG loss '
                       '(no_vgg_loss + skip_unused_losses: VGG is not run at all)')
  ap.add_argument('--no-cpu-baseline', action='store_true')
  ap.add_argument('--no-kernel-timers', action='store_true',
                  help='A/B only: leave the in-library hipEvent timers off in the timed region (no roofline objects)')
  ap.add_argument('--late-readback', action='store_true',
                  help='A/B only: read the losses back after the optimizer steps, as round 2 did')
  ap.add_argument('--unfused-d-lrelu', action='store_true',
                  help="A/B only: the PatchGAN layer-0 LeakyReLU backward as its own pass instead of in layer 1's epilogue")
  ap.add_argument('--unfused-norm-sums', action='store_true',
                  help='A/B only: every InstanceNorm backward computes its two sums in its own pass instead of taking them from the '
                       'epilogue of the data-gradient kernel that produced dy (round 4)')
  ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                  help='collective backend for N > 1: nccl == RCCL over xGMI (the product path); gloo only to rehearse '
                       'the multi-rank schedule on a box with fewer GPUs than ranks (with --share-gpu)')
  ap.add_argument('--share-gpu', action='store_true',
                  help='rehearsal: ranks take device LOCAL_RANK %% device_count (several ranks per GPU; needs --backend gloo)')
  ap.add_argument('--grad-reduce', default='auto', choices=['auto', 'bf16', 'fp32'],
                  help='N > 1: wire dtype of the gradient all-reduce.  auto (default) = bf16 -- half the xGMI bytes, one rounding '
                       'of each rank\'s contribution and of the sum -- and the SAME run then repeats the timed steps on an fp32 '
                       'wire and prints that figure next to it as "fp32_wire" (SURVEY.md 8d config 4 asks for both)')
  ap.add_argument('--bf16-reduce', action='store_true', help='same as --grad-reduce bf16 (kept for round-2/3 command lines)')
  ap.add_argument('--ddp-overlap', default='d_backward', choices=['d_backward', 'layers'],
                  help='N > 1: the generator gradient all-reduce overlaps the discriminator backward (default) or is fired '
                       'layer by layer from the backward hooks (round 1-2 behaviour); same results')
  ap.add_argument('--master-port', type=int, default=0, help='self-launch only: rendezvous port (0 = pick a free one)')
  ap.add_argument('--debug-mode', type=int, default=None,
                  help='developer A/B runs: use libjpdse_hip_dev.so with this kernel-selection mode (include/jpdse_dev.h)')
  return ap.parse_args()


def make_opt(args, device_index):
  from ctu.utils.synthetic import default_opt
  kw = dict(gpu_ids=[device_index], print_losses=False, compute_dtype=args.dtype, use_compressed=True,
            netG=args.netG, ngf=64 if args.netG == 'global' else 32, batch_size=args.batch,
            bf16_grad_reduce=(args.bf16_reduce or args.grad_reduce in ('auto', 'bf16')), ddp_overlap=args.ddp_overlap)
  if args.no_vgg:
    kw.update(no_vgg_loss=True, skip_unused_losses=True)
  return default_opt(**kw)


def cpu_baseline(args, budget_s=75.0):
  """The oracle (a port: kind 'port') on this box's host cores, SURVEY.md 8(d): fp32, batch 1.  A batch-1 conv graph does
  not scale to every hardware thread of a large host (round 2 ran it on all 128 and got LESS than the survey's 8-core probe),
  so the thread count is swept first -- one 512x256 step each at {8, 16, 32, 64, all} threads after a warm-up step, stopping once a
  count is 1.5x slower than the best so far -- and the fastest is kept; with it, one 1024x512 step (the headline size) is timed when the projection fits the budget, else the
  512x256 figure is quoted.  `sample` says which, and which thread count won.  Bounded: about `budget_s` seconds."""
  from oracle.ctu_cpu import model as omodel
  try:
    ncpu = len(os.sched_getaffinity(0))          # the cores this process may use, not the host's (a 16-core share of 256)
  except (AttributeError, OSError):
    ncpu = os.cpu_count() or torch.get_num_threads()
  opt = omodel.default_opt(netG=args.netG, ngf=64 if args.netG == 'global' else 32, use_compressed=True)
  torch.manual_seed(1234)
  ora = omodel.OracleTrainer(opt)

  def one(h, w, seed):
    xd = omodel.synthetic_batch(1, h, w, seed=seed)
    t0 = time.perf_counter()
    ora.step(xd)
    return time.perf_counter() - t0

  t_start = time.perf_counter()
  saved = torch.get_num_threads()
  cands = sorted({c for c in (8, 16, 32, 64, ncpu) if 0 < c <= ncpu})
  torch.set_num_threads(cands[0])
  one(256, 512, 1)                                    # warm-up (allocator, oneDNN primitive cache)
  sweep = {}
  for i, c in enumerate(cands):
    if sweep and time.perf_counter() - t_start > 0.6 * budget_s:
      break
    if sweep and sweep[max(sweep)] > 1.5 * min(sweep.values()):
      break                                     # already past the knee: more threads only oversubscribe (round 3: 256 threads took
                                                # 214 s for the step that 8 threads do in 1.3 s)
    torch.set_num_threads(c)
    sweep[c] = one(256, 512, 2 + i)
  best = min(sweep, key=sweep.get)
  torch.set_num_threads(best)
  sweep_txt = ', '.join('%d: %.2f s' % (c, t) for c, t in sorted(sweep.items()))
  # the quoted figure is the MEDIAN of three timed steps at the chosen thread count (the sweep's single samples only
  # pick the count), at 1024x512 -- the headline size -- when three such steps fit the budget, else at 512x256
  med = lambda v: sorted(v)[len(v) // 2]
  if (time.perf_counter() - t_start) + 3 * 4.3 * sweep[best] <= budget_s:
    ts = [one(512, 1024, 11 + i) for i in range(3)]
    size = '1024x512'
  else:
    ts = [one(256, 512, 21 + i) for i in range(3)]
    size = '512x256'
  value = 1.0 / med(ts)
  sample = ('median of 3 steps at %s (%s s) on %d threads -- the fastest count of a sweep of one 512x256 step per thread count '
            '{%s} after a warm-up step; %d cores available to this process; batch 1, fp32 torch-CPU oracle (the GPU figure is '
            'batch %d)' % (size, ' / '.join('%.2f' % t for t in ts), best, sweep_txt, ncpu, args.batch))
  torch.set_num_threads(saved)
  return dict(value=round(value, 5), unit='images/sec', cores=best, cores_available=ncpu, kind='port', sample=sample)


def visible_gpu_count():
  """GPUs this process tree may use, WITHOUT a HIP / torch.cuda call in this (launcher) process: the GPU nodes of the KFD
  topology (a node with simd_count > 0), cut down by the *_VISIBLE_DEVICES masks the children will inherit.  None when
  neither source says anything (no readable topology, no mask): the ranks then find out for themselves."""
  n = None
  base = '/sys/class/kfd/kfd/topology/nodes'
  try:
    count = 0
    for node in sorted(os.listdir(base)):
      try:
        with open(os.path.join(base, node, 'properties')) as fh:
          props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        if int(props.get('simd_count', '0')) > 0:
          count += 1
      except (OSError, ValueError):
        continue
    n = count
  except OSError:
    n = None
  for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
    if var in os.environ:
      ids = [t for t in os.environ[var].split(',') if t.strip() != '']
      n = len(ids) if n is None else min(n, len(ids))
  return n


def self_launch(args, timeout_s=1500):
  """`python bench.py --gpus N` (N > 1) without a launcher: start `torch.distributed.run` with N ranks as a CHILD process
  and relay its output.  This process never touches the GPU (argparse, `import torch`, sysfs only -- no torch.cuda call)
  and never execs.  A child that hangs is killed with its process group after `timeout_s`."""
  import signal
  import socket
  import subprocess
  port = args.master_port
  if port == 0:
    with socket.socket() as sk:
      sk.bind(('127.0.0.1', 0))
      port = sk.getsockname()[1]
  ndev = visible_gpu_count()
  if ndev is not None and ndev < args.gpus and not args.share_gpu:
    sys.stderr.write('bench.py: --gpus %d but %d GPU(s) visible (rehearse with --backend gloo --share-gpu)\n' % (args.gpus, ndev))
    return 2
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC only on this pool (RCCL across processes)
  env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
         '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
  proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)

  def _forward_kill(signum, frame):           # this launcher is being terminated (an outer `timeout`, the driver): the ranks are in
    try:                                      # their own session and would outlive it -- take the exact group this launcher started
      os.killpg(proc.pid, signal.SIGKILL)
    except OSError:
      pass
    sys.exit(128 + signum)

  for sig in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
    signal.signal(sig, _forward_kill)
  try:
    out, _ = proc.communicate(timeout=timeout_s)
  except subprocess.TimeoutExpired:
    try:
      os.killpg(proc.pid, signal.SIGKILL)      # the exact group this launcher started
    except OSError:
      pass
    out, _ = proc.communicate()
    sys.stdout.write(out or '')
    sys.stderr.write('bench.py: the %d-rank child did not finish within %d s and was killed\n' % (args.gpus, timeout_s))
    return 3
  sys.stdout.write(out)
  sys.stdout.flush()
  return proc.returncode


def main():
  args = parse()
  if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
    sys.exit(self_launch(args))
  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    sys.stderr.write('bench.py: WORLD_SIZE=%d but --gpus %d (launch with --nproc-per-node %d)\n' % (world, args.gpus, args.gpus))
    sys.exit(2)
  dev_index = 0
  if world > 1:
    dev_index = local_rank % torch.cuda.device_count() if args.share_gpu else local_rank
    torch.cuda.set_device(dev_index)
    if args.backend == 'nccl':
      dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', dev_index))
    else:
      dist.init_process_group('gloo', rank=rank, world_size=world)
  else:
    torch.cuda.set_device(0)
  dev = torch.device('cuda', dev_index)

  import jpdse_hip
  from jpdse_hip import lib, check
  from ctu.trainers import get_trainer
  from ctu.utils.synthetic import synthetic_batch
  jpdse_hip.require_gpu(dev_index)
  if args.debug_mode is not None:      # developer A/B runs only: the whole bench on libjpdse_hip_dev.so in that mode
    jpdse_hip.set_dev_mode(args.debug_mode)

  torch.manual_seed(1234)            # identical replicas; enable_data_parallel also broadcasts rank 0
  opt = make_opt(args, dev_index)
  import contextlib
  with contextlib.redirect_stdout(sys.stderr):       # the reference prints a banner here; stdout carries ONE JSON line
    trainer = get_trainer(opt)(opt, 'train')

  # synthetic Cityscapes-shaped batch, seeded per rank, resident in HBM before the clock starts
  xd = synthetic_batch(args.batch, args.height, args.width, seed=1234 + rank)
  xd = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in xd.items()}

  verbose = os.environ.get('JPDSE_BENCH_VERBOSE') == '1'

  def note(msg):                      # progress on stderr (JPDSE_BENCH_VERBOSE=1): where a multi-rank run stands
    if verbose:
      sys.stderr.write('[bench rank %d] %s\n' % (rank, msg))
      sys.stderr.flush()

  def barrier():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  # A full (generation-2) Python GC pass over the freshly built module tree takes ~65 ms and would fire once
  # around step 20 (scripts/diag_step_times.py): collect now and move the survivors out of the collector's
  # way, as a long-running training loop would.  No work of the step is skipped.
  import gc
  gc.collect()
  gc.freeze()
  note('trainer built, %d warm-up steps' % args.warmup)
  for i in range(args.warmup):
    trainer.step(xd)
    if verbose:
      torch.cuda.synchronize()
      note('warm-up step %d done' % i)
  # time the ResnetBlock 3x3 GEMM (forward and its data-gradient: same kernel, N=1024, K=9216)
  L = lib()
  if args.late_readback:
    trainer.model.early_loss_readback = False
  if args.unfused_norm_sums:
    from jpdse_hip.layers import HipResnetBlock
    HipResnetBlock.fuse_norm_sums = False
  if args.unfused_d_lrelu:
    for m in trainer.model.netD.modules():
      if hasattr(m, 'fuse_lrelu0'):
        m.fuse_lrelu0 = False
  # In-library kernel timers (hipEvent pairs on the kernels' own stream, inside the timed region).  Every event costs the
  # stream ~3.7 us (measured round 3: 296 events per step = 1.1 ms of a 26.3 ms step), so they cover the FIRST
  # step of the timed region only -- one step holds 36 ResnetBlock GEMM launches, 18 weight gradients, 72 norm calls.
  # Round 4: the ResnetBlock GEMM regions -- the headline `roofline` fractions -- are sampled on the first
  # THREE timed steps (3 x 72 regions = 432 events = 1.6 ms spread over the timed region: 0.08 ms per step at the driver's 20
  # steps), the HBM regions on the first step (148 events = 0.55 ms); `kernel_timers` in the output says so.
  timed_steps = 0 if args.no_kernel_timers else (min(3, args.steps) if args.steps >= 12 else min(1, args.steps))   # short runs: one step (0.5 ms of events)
  hbm_steps = 0 if args.no_kernel_timers else min(1, args.steps)
  tsteps = max(timed_steps, 1)
  hsteps = max(hbm_steps, 1)
  if timed_steps:
    check(L.jpdse_prof_select(1, 1024, 9216, 72 * timed_steps), 'prof_select')        # 36 halo + 18 ring + 18 wgrad regions per step
    check(L.jpdse_prof_hbm_select(1, 74 * hbm_steps), 'prof_hbm_select')             # 72 norm calls + 2 Adams per step
  if world > 1:
    trainer.model.ddp_timeline = []          # per-step timing events of the data-parallel schedule (7 events per step)
  barrier()
  note('timed region: %d steps' % args.steps)
  t0 = time.perf_counter()
  for _ in range(args.steps):
    trainer.step(xd)
  barrier()
  elapsed = time.perf_counter() - t0
  note('timed region done: %.1f ms per step' % (1e3 * elapsed / args.steps))
  ddp_info = None
  if world > 1:
    tl, trainer.model.ddp_timeline = trainer.model.ddp_timeline, None
    mean = lambda a, b: sum(t[a].elapsed_time(t[b]) for t in tl) / max(len(tl), 1)
    bg = trainer.model.grad_buckets.get('G')
    ddp_info = dict(overlap=args.ddp_overlap, g_buckets=len(bg.buckets) if bg is not None else 0,
                    g_bytes=bg.total_bytes() if bg is not None else 0,
                    d_backward_ms=round(mean('g_bwd_end', 'd_bwd_end'), 3),
                    g_allreduce_exposed_ms=round(mean('d_bwd_end', 'g_reduced'), 3),
                    d_allreduce_exposed_ms=round(mean('adam_g_end', 'd_reduced'), 3),
                    note='compute-stream events; exposed = what the compute stream waited for the collective beyond the work '
                         'it overlaps (DESIGN.md 6 budget: <= 1.5 ms per step at 8 GPUs on the bf16 wire)')
  # N > 1, --grad-reduce auto: the same K steps again on an fp32 wire (twice the xGMI bytes), reported next to the bf16 figure
  fp32_wire = None
  if world > 1 and args.grad_reduce == 'auto' and not args.bf16_reduce:
    try:
      trainer.enable_data_parallel(reduce_dtype=None)
      for _ in range(max(1, min(2, args.warmup))):
        trainer.step(xd)
      barrier()
      t1 = time.perf_counter()
      for _ in range(args.steps):
        trainer.step(xd)
      barrier()
      e2 = time.perf_counter() - t1
      t = torch.tensor([e2], dtype=torch.float64, device=dev)
      dist.all_reduce(t, op=dist.ReduceOp.MAX)
      e2 = float(t.item())
      fp32_wire = dict(value=round(args.batch * world * args.steps / e2, 4), ms_per_step=round(1e3 * e2 / args.steps, 3),
                       steps=args.steps, note='same process, same weights, gradient all-reduce on an fp32 wire')
    except Exception as e:             # the headline (bf16-wire) figure above must survive a failure of the side measurement
      fp32_wire = dict(error=repr(e)[:300])
  ms, fl, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
  extra = {}
  for cls, name in ((1, 'ring'), (2, 'wgrad')):      # the other two regions of the same layer, before the log is reset
    m2, f2, n2 = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    check(L.jpdse_prof_collect_class(cls, ctypes.byref(m2), ctypes.byref(f2), ctypes.byref(n2)), 'prof_collect_class')
    extra[name] = (m2.value, f2.value, n2.value)
  check(L.jpdse_prof_collect(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(n)), 'prof_collect')
  check(L.jpdse_prof_select(0, 0, 0, 0), 'prof_select off')
  hbm = {}
  for cls, name in ((0, 'inorm_fwd'), (1, 'inorm_bwd'), (2, 'adam')):
    m2, b2, n2 = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    check(L.jpdse_prof_hbm_collect(cls, ctypes.byref(m2), ctypes.byref(b2), ctypes.byref(n2)), 'prof_hbm_collect')
    hbm[name] = (m2.value, b2.value, n2.value)
  check(L.jpdse_prof_hbm_select(0, 0), 'prof_hbm_select off')
  if world > 1:
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  if rank == 0:
    images = args.batch * world * args.steps
    value = images / elapsed
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    roof = None
    if n.value > 0 and ms.value > 0:
      achieved = fl.value / (ms.value * 1e-3) / 1e12
      roof = dict(bound='mfma', kernel='gemm_halo_kernel (ResnetBlock 3x3 conv fwd + data gradient, N=1024 K=9216)',
                  achieved=round(achieved, 2), peak=peak, unit='TFLOP/s', frac=round(achieved / peak, 4),
                  traffic=None, launches_per_step=n.value / tsteps,
                  avg_launch_ms=round(ms.value / n.value, 4),
                  flops_per_launch=fl.value / n.value)
      # HBM-side bytes per launch of the same kernel: rocprofv3 PMC counters cannot be read from inside this
      # process, so the figure is the committed one of scripts/run_pmc_hbm.sh (separate --pmc passes over this
      # very command, FETCH_SIZE doubled per the gfx950 correction) -- only quoted for the workload it was taken on
      tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
      if (os.path.exists(tpath) and (args.netG, args.width, args.height, args.batch, args.dtype) ==
          ('global', 1024, 512, 4, 'bf16')):
        with open(tpath) as fh:
          tj = json.load(fh)
        roof['traffic'] = tj['fetch_bytes_per_launch'] + tj['write_bytes_per_launch']
        roof['traffic_unit'] = 'bytes per launch (memory-side of L2, Infinity-Cache hits included)'
        roof['traffic_source'] = tj['source']
    # second roofline entry: ALL THREE passes of the ResnetBlock 3x3 convs -- forward, data gradient including the ring
    # strips / fold of the reflect padding, weight gradient -- so that ">= 40 % on the 3x3 convs" is judged on the whole
    # layer and not on its best kernel
    roof_all = None
    if roof is not None and extra['wgrad'][2] > 0:
      t_all = ms.value + extra['ring'][0] + extra['wgrad'][0]
      f_all = fl.value + extra['wgrad'][1]
      ach = f_all / (t_all * 1e-3) / 1e12
      wg = extra['wgrad']
      roof_all = dict(bound='mfma', kernel='ResnetBlock 3x3 conv, all passes: gemm_halo_kernel (fwd, dgrad over the folded frame) + '
                                          'ring_frame_kernel (the "ring" time) + wgrad_nine_kernel',
                      achieved=round(ach, 2), peak=peak, unit='TFLOP/s', frac=round(ach / peak, 4),
                      ms_per_step=dict(fwd_dgrad=round(ms.value / tsteps, 4), ring=round(extra['ring'][0] / tsteps, 4),
                                       wgrad=round(wg[0] / tsteps, 4)),
                      wgrad=dict(achieved=round(wg[1] / (wg[0] * 1e-3) / 1e12, 2), avg_launch_ms=round(wg[0] / wg[2], 4),
                                 launches_per_step=wg[2] / tsteps, frac=round(wg[1] / (wg[0] * 1e-3) / 1e12 / peak, 4)))
    # third roofline entry, HBM-bound: the InstanceNorm + activation (+ residual) calls, forward and backward (north_star:
    # "HBM GB/s on the norm/activation kernels against gfx950 peak"), and the fused Adam next to them
    roof_hbm = None
    nt = hbm['inorm_fwd'][0] + hbm['inorm_bwd'][0]
    if nt > 0:
      nb = hbm['inorm_fwd'][1] + hbm['inorm_bwd'][1]
      gbs = lambda b, t: round(b / (t * 1e-3) / 1e9, 1) if t > 0 else None
      roof_hbm = dict(bound='hbm', kernel='InstanceNorm + activation (+ residual): moment / finalize / apply and register-held kernels, '
                                           'forward and backward (norm.hip)',
                      achieved=gbs(nb, nt), peak=HBM_PEAK_GBS, unit='GB/s', frac=round(nb / (nt * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      traffic=None, calls_per_step=(hbm['inorm_fwd'][2] + hbm['inorm_bwd'][2]) / hsteps,
                      ms_per_step=round(nt / hsteps, 4), algorithmic_bytes_per_step=nb / hsteps,
                      forward=dict(achieved=gbs(hbm['inorm_fwd'][1], hbm['inorm_fwd'][0]), ms_per_step=round(hbm['inorm_fwd'][0] / hsteps, 4)),
                      backward=dict(achieved=gbs(hbm['inorm_bwd'][1], hbm['inorm_bwd'][0]), ms_per_step=round(hbm['inorm_bwd'][0] / hsteps, 4)),
                      adam=dict(achieved=gbs(hbm['adam'][1], hbm['adam'][0]), ms_per_step=round(hbm['adam'][0] / hsteps, 4),
                                frac=round(hbm['adam'][1] / (hbm['adam'][0] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if hbm['adam'][0] > 0 else None))
      tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
      if (os.path.exists(tpath) and (args.netG, args.width, args.height, args.batch, args.dtype) == ('global', 1024, 512, 4, 'bf16')):
        with open(tpath) as fh:
          tj = json.load(fh)
        if 'norm_family_bytes_per_step' in tj:
          roof_hbm['traffic'] = tj['norm_family_bytes_per_step']
          roof_hbm['traffic_unit'] = 'bytes per step, all norm kernels (memory-side of L2, Infinity-Cache hits included)'
          roof_hbm['traffic_source'] = tj['source']
    f_alg = F_ALG_GFLOP.get((args.netG, args.width, args.height))
    if args.no_vgg and f_alg:      # minus VGG's 3 passes (2 fwd + 1 dgrad), SURVEY.md 8d config 2
      f_alg = round(f_alg - 6 * 189.4 * (args.width * args.height) / (1024.0 * 512.0), 1)
    out = {
        'metric': 'training images/sec at 1024x512 (G+D+VGG step)' if (args.width, args.height) == (1024, 512)
                  else 'training images/sec at %dx%d (G+D+VGG step)' % (args.width, args.height),
        'value': round(value, 4), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
        'data': 'synthetic',
        'config': {'workload': 'JPD-SE train step %dx%d, netG=%s ngf=%d (39-ch input), num_D=2, %sfeat+L1, '
                               'Adam x2, batch %d/GPU' % (args.width, args.height, args.netG, opt.ngf,
                                                          '' if args.no_vgg else 'VGG19+', args.batch),
                   'global_batch': args.batch * world, 'parallelism': 'dp%d' % world,
                   'grad_reduce': ('bf16' if (args.bf16_reduce or args.grad_reduce in ('auto', 'bf16')) else 'fp32') if world > 1 else None,
                   'step_mfma_frac': (round(f_alg * value / 1e3 / peak, 4) if f_alg else None),
                   'f_alg_gflop_per_image': f_alg},
        'roofline': roof,
        'roofline_resblock_all_passes': roof_all,
        'roofline_hbm': roof_hbm,
        'kernel_timers': {'steps_covered': timed_steps, 'hbm_steps_covered': hbm_steps, 'of_timed_steps': args.steps,
                          'note': 'hipEvent pairs inside the timed region: ResnetBlock GEMM regions on the first 3 steps (1 when fewer '
                                  'than 12 steps are timed), norm / Adam regions on the first (each event costs the stream ~3.7 us: '
                                  '~2.1 ms in all at 3 steps)'},
        'multi_gpu': ({'ddp': ddp_info, 'fp32_wire': fp32_wire} if world > 1 else
                      'N = 1: no collective ran; the 1/2/4/8-GPU curve needs the driver\'s 8-GPU node (python bench.py --gpus N)'),
    }
    if world == 1 and not args.no_cpu_baseline:
      out['cpu_baseline'] = cpu_baseline(args)
    else:
      out['cpu_baseline'] = None
    print(json.dumps(out))
  if world > 1:
    dist.destroy_process_group()


if __name__ == '__main__':
  main()
