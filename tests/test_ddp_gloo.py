"""CPU, world_size 2 over gloo: the data-parallel exchange step (jpdse_hip.ddp.GradBuckets) and
the claim it rests on: per-rank gradients of equal image shards, SUM-all-reduced
and scaled by 1/world, equal the single-process gradient of the concatenated batch, because
InstanceNorm is per sample and every loss is a batch mean."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q, wire='fp64'):
  for p in (ROOT, os.path.join(ROOT, 'jpd-se_amd')):
    if p not in sys.path:
      sys.path.insert(0, p)
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    torch.set_num_threads(2)
    from jpdse_hip.ddp import GradBuckets
    from oracle.ctu_cpu import model as omodel
    kw = dict(ngf=4, ndf=4, n_blocks_global=1)
    torch.manual_seed(7)                       # identical replicas
    ora = omodel.OracleTrainer(omodel.default_opt(**kw))
    full = omodel.synthetic_batch(2 * world, 32, 64, seed=3)
    shard = {k: (v[2 * rank: 2 * rank + 2] if torch.is_tensor(v) else v) for k, v in full.items()}
    gG, gD = ora.grads_in_dtype(shard, torch.float64)     # this rank's shard
    # parameters in channels_last like the product's, gradients re-homed into flat buckets
    params = []
    pdt = torch.float64 if wire == 'fp64' else torch.float32
    for k, v in ora.G.items():
      p = torch.nn.Parameter(v.detach().to(pdt).clone())
      if p.dim() == 4:
        p.data = p.data.contiguous(memory_format=torch.channels_last)
      params.append((k, p))
    buckets = GradBuckets(params, bucket_bytes=64 << 10,   # small buckets: several per network
                          reduce_dtype=torch.bfloat16 if wire == 'bf16' else None)
    assert len(buckets.buckets) > 2
    for k, p in reversed(params):                          # backward order
      p.grad.copy_(gG[k])
      buckets.mark_ready(p)
    buckets.finish()
    avg = {k: p.grad.clone() / world for k, p in params}
    if rank == 0:
      refG, _ = ora.grads_in_dtype(full, torch.float64)    # single process, global batch
      worst = 0.0
      for k, _ in params:
        if k.endswith('.weight'):
          worst = max(worst, ((avg[k] - refG[k]).abs().max() / refG[k].abs().max()).item())
      q.put(('ok', worst, len(buckets.buckets), buckets.total_bytes()))
  except Exception as e:   # surface the failure in the parent
    if rank == 0:
      q.put(('error', repr(e), 0, 0))
    raise
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('wire', ['fp64', 'bf16'])
def test_two_rank_gradient_average_equals_global_batch(wire):
  """wire = bf16: the optional half-width all-reduce: fp32 buckets cast to a bf16 wire buffer,
  summed over the ranks, cast back -- the average then carries bf16 rounding (2^-8 relative per element)."""
  ctx = mp.get_context('spawn')
  q = ctx.Queue()
  port = 29500 + (os.getpid() % 2000) + (17 if wire == 'bf16' else 0)
  procs = [ctx.Process(target=_worker, args=(r, 2, port, q, wire)) for r in range(2)]
  for p in procs:
    p.start()
  status, worst, n_buckets, nbytes = q.get(timeout=600)
  for p in procs:
    p.join(timeout=120)
  assert status == 'ok', worst
  assert all(p.exitcode == 0 for p in procs)
  if wire == 'fp64':
    assert worst < 1e-9, worst       # fp64: only summation order differs
  else:
    assert 1e-6 < worst < 1.5e-2, worst   # bf16 wire: rounding of each rank's contribution and of the sum
  assert n_buckets > 2 and nbytes > 0


def test_buckets_single_process_views_and_order():
  """Without a process group the buckets are pure bookkeeping: grads are views into the flat
  buffers, laid out in reverse-forward order with 16-byte aligned starts."""
  sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
  from jpdse_hip.ddp import GradBuckets
  ps = [('a.weight', torch.nn.Parameter(torch.randn(4, 3, 3, 3).contiguous(memory_format=torch.channels_last))),
        ('a.bias', torch.nn.Parameter(torch.randn(5))),
        ('b.weight', torch.nn.Parameter(torch.randn(7, 4, 3, 3).contiguous(memory_format=torch.channels_last)))]
  for _, p in ps:
    p.grad = torch.ones_like(p, memory_format=torch.preserve_format)
  b = GradBuckets(ps, bucket_bytes=1 << 20)
  assert len(b.buckets) == 1
  flat = b.buckets[0]['flat']
  names = [n for n, _, _, _ in b.buckets[0]['params']]
  assert names == ['b.weight', 'a.bias', 'a.weight']
  for n, p, off, numel in b.buckets[0]['params']:
    assert off % 4 == 0 and p.grad.stride() == p.stride()
    assert p.grad.data_ptr() == flat.data_ptr() + flat.element_size() * off
    assert float(p.grad.sum()) == numel          # previous gradient values were carried over
  ps[0][1].grad.fill_(2.0)
  assert float(flat.sum()) == 2 * ps[0][1].numel() + ps[1][1].numel() + ps[2][1].numel()
  for _, p in ps:
    b.mark_ready(p)
  b.finish()      # world size 1: no collective, just resets the counters
  assert all(x['pending'] == len(x['params']) for x in b.buckets)


def test_world_size_one_group_still_runs_the_collective():
  """always_reduce (what the trainer's data-parallel mode sets): with a process group of ONE rank the buckets still go
  through all_reduce + wait -- the way the RCCL path is exercised on a one-GPU box (tests/test_hip_ddp.py)."""
  sys.path.insert(0, os.path.join(ROOT, 'jpd-se_amd'))
  from jpdse_hip.ddp import GradBuckets
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29900 + os.getpid() % 90))
  dist.init_process_group('gloo', rank=0, world_size=1)
  try:
    ps = [('w', torch.nn.Parameter(torch.randn(4, 3, 3, 3).contiguous(memory_format=torch.channels_last))),
          ('b', torch.nn.Parameter(torch.randn(5)))]
    for _, p in ps:
      p.grad = torch.randn_like(p, memory_format=torch.preserve_format)
    want = [p.grad.clone() for _, p in ps]
    plain = GradBuckets(ps, bucket_bytes=1 << 20)
    forced = GradBuckets(ps, bucket_bytes=1 << 20, always_reduce=True, reduce_dtype=torch.bfloat16)
    for _, p in ps:
      plain.mark_ready(p)
    assert plain.buckets[0]['handle'] is None           # world size 1, not forced: no collective
    plain.finish()
    for _, p in ps:
      forced.mark_ready(p)
    assert forced.buckets[0]['handle'] is not None and forced.buckets[0]['wire'].dtype == torch.bfloat16
    forced.finish()
    for (_, p), w in zip(ps, want):
      assert torch.equal(p.grad, w.to(torch.bfloat16).to(torch.float32))   # through the bf16 wire and back
  finally:
    dist.destroy_process_group()
