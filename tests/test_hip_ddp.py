"""GPU, 2 ranks sharing cuda:0 (gloo carries the collective; RCCL needs one device per rank and
the driver runs the real multi-GPU bench): the trainer's data-parallel path end to end --
weight broadcast, gradients re-homed into flat bucket views, async all-reduce fired from the
backward hooks, 1/world folded into fused Adam -- against ONE oracle step on the global batch."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
  for p in (ROOT, os.path.join(ROOT, 'jpd-se_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
      sys.path.insert(0, p)
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    torch.cuda.set_device(0)
    from ctu.trainers import get_trainer
    from oracle.ctu_cpu import model as omodel
    kw = dict(ngf=8, ndf=8, n_blocks_global=1)
    torch.manual_seed(1234)
    ora = omodel.OracleTrainer(omodel.default_opt(**kw))
    opt = omodel.default_opt(gpu_ids=[0], print_losses=False, **kw)
    torch.manual_seed(100 + rank)               # replicas start DIFFERENT: the broadcast must fix that
    tr = get_trainer(opt)(opt, 'train')
    if rank == 0:
      tr.model.netG.load_state_dict({k: v.detach() for k, v in ora.G.items()})
      tr.model.netD.load_state_dict({k: v.detach() for k, v in ora.D.items()})
    tr.enable_data_parallel(bucket_bytes=32 << 10)
    assert tr.optimizer_G.grad_scale == 0.5 and len(tr.model.grad_buckets['G'].buckets) > 1
    full = omodel.synthetic_batch(2 * world, 32, 64, seed=5)
    shard = {k: (v[2 * rank: 2 * rank + 2] if torch.is_tensor(v) else v) for k, v in full.items()}
    tr.step(shard)
    torch.cuda.synchronize()
    # every rank must hold the same weights after the step
    flatw = torch.cat([p.detach().float().reshape(-1) for p in tr.model.netG.parameters()]).cpu()
    gathered = [torch.zeros_like(flatw) for _ in range(world)]
    dist.all_gather(gathered, flatw)
    if rank == 0:
      same = all(torch.equal(gathered[0], g) for g in gathered)
      ora.step(full)
      worst = 0.0
      for net, ref in ((tr.model.netG, ora.G), (tr.model.netD, ora.D)):
        for k, v in net.state_dict().items():
          if k.endswith('.weight'):
            a, b = v.cpu().double(), ref[k].detach().double()
            worst = max(worst, ((a - b).norm() / b.norm()).item())
      q.put(('ok', same, worst))
  except Exception as e:
    if rank == 0:
      q.put(('error', repr(e), 0.0))
    raise
  finally:
    dist.destroy_process_group()


def test_two_rank_step_equals_global_batch_step():
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29600 + (os.getpid() % 2000)
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
  for p in procs:
    p.start()
  status, same, worst = q.get(timeout=900)
  for p in procs:
    p.join(timeout=120)
  assert status == 'ok', same
  assert same, 'replicas diverged after one data-parallel step'
  assert worst <= 3e-3, 'weights after the 2-rank step differ from the global-batch oracle step: %.3e' % worst


def _nccl_single_rank_worker(q, bf16_wire):
  for p in (ROOT, os.path.join(ROOT, 'jpd-se_amd'), os.path.join(ROOT, 'tests')):
    if p not in sys.path:
      sys.path.insert(0, p)
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29700 + os.getpid() % 200))
  os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  torch.cuda.set_device(0)
  dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
  try:
    from ctu.trainers import get_trainer
    from oracle.ctu_cpu import model as omodel
    kw = dict(ngf=16, ndf=16, n_blocks_global=2)
    torch.manual_seed(1234)
    ora = omodel.OracleTrainer(omodel.default_opt(**kw))
    opt = omodel.default_opt(gpu_ids=[0], print_losses=False, **kw)

    def make():
      tr = get_trainer(opt)(opt, 'train')
      tr.model.netG.load_state_dict({k: v.detach() for k, v in ora.G.items()})
      tr.model.netD.load_state_dict({k: v.detach() for k, v in ora.D.items()})
      return tr

    plain, dp = make(), make()
    assert not plain.model.grad_buckets                       # world size 1: data parallelism is opt-in
    dp.enable_data_parallel(bucket_bytes=256 << 10, reduce_dtype=torch.bfloat16 if bf16_wire else None)
    bg = dp.model.grad_buckets['G']
    assert len(bg.buckets) > 2 and dp.optimizer_G.grad_scale == 1.0
    launched = []
    orig = bg._launch
    bg._launch = lambda b: (orig(b), launched.append(b['handle'] is not None))[0]
    res = []
    for s in range(2):
      xd = omodel.synthetic_batch(2, 64, 128, seed=5 + s)
      a, b = plain.step(xd), dp.step(xd)
      torch.cuda.synchronize()
      res.append((a, b, dict(plain.last_losses), dict(dp.last_losses)))
    worst, equal = 0.0, True
    for (k, v), (_, w) in zip(plain.model.netG.state_dict().items(), dp.model.netG.state_dict().items()):
      worst = max(worst, ((v.double() - w.double()).norm() / v.double().norm().clamp_min(1e-30)).item())
      equal = equal and torch.equal(v, w)
    # every .grad of the data-parallel trainer lives inside a bucket's flat buffer (RCCL reduced it in place)
    inside = True
    for b in bg.buckets:
      lo, hi = b['flat'].data_ptr(), b['flat'].data_ptr() + b['flat'].numel() * 4
      inside = inside and all(lo <= p.grad.data_ptr() < hi for _, p, _, _ in b['params'])
    q.put(('ok', res, worst, equal, all(launched) and len(launched) >= len(bg.buckets), inside))
  except Exception as e:
    q.put(('error', repr(e), 0.0, False, False, False))
    raise
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('bf16_wire', [False, True], ids=['fp32_wire', 'bf16_wire'])
def test_rccl_world_size_one_step_equals_plain_step(bf16_wire):
  """The REAL RCCL path on one GPU: a world-size-1 `nccl` process group, enable_data_parallel forced on, so every bucket
  goes through ncclAllReduce in place on the strided-view gradient memory, launched asynchronously from the backward hooks,
  and the compute stream waits on the work handles before Adam.  With one rank the sum is the identity, so the step must
  equal the plain (bucket-less) step: bit-for-bit on an fp32 wire; through the optional bf16 wire the gradients are
  rounded to bf16 once: Adam then flips the +-lr update of a few near-zero-gradient elements, nothing more."""
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  p = ctx.Process(target=_nccl_single_rank_worker, args=(q, bf16_wire))
  p.start()
  status, res, worst, equal, launched, inside = q.get(timeout=900)
  p.join(timeout=120)
  assert status == 'ok', res
  assert p.exitcode == 0
  assert launched, 'a bucket was not handed to the collective'
  assert inside, 'a gradient is not a view into its all-reduce bucket'
  for a, b, La, Lb in res[:1]:
    assert a == b and La == Lb, 'step 0: identical weights must give identical losses'
  if bf16_wire:
    assert worst <= 5e-3, worst
  else:
    assert equal, 'fp32 wire, one rank: the data-parallel step must be bit-identical to the plain step (worst %.3e)' % worst
