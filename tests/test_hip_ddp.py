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
