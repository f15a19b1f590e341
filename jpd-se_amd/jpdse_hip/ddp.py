"""Per-image data parallelism: bucketed gradient all-reduce over torch.distributed
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference refuses multi-GPU outright (ctu/parsers/base_parser.py:234-237); this is the
new capability SURVEY.md §8e specifies: one process per GPU, full replicas, images sharded,
ONE exchange step per optimizer -- an all-reduce(SUM) of that network's gradients, averaged
by folding 1/world_size into the fused Adam kernel's grad_scale.

Zero-copy design for 288 GB HBM: every bucket is one persistent flat fp32 buffer and each
parameter's `.grad` is a strided view into it, so the wgrad kernels write straight into
communication memory and RCCL reduces in place.  Buckets follow reverse-backward order
(last layers first) and are launched asynchronously the moment their last gradient has been
written, so the 18 ResnetBlock filters (93 % of the bytes) are on the wire while the long
full-resolution layers of the backward are still computing.
"""
import torch
import torch.distributed as dist


class GradBuckets(object):

  def __init__(self, named_params, bucket_bytes=64 << 20, process_group=None, reverse=True, reduce_dtype=None,
               always_reduce=False, defer=False):
    """named_params: iterable of (name, param) in FORWARD order.  reduce_dtype=torch.bfloat16: the bucket is cast
    to a bf16 wire buffer before the all-reduce and back afterwards (half the xGMI bytes; the sum then carries bf16
    rounding).  always_reduce: issue the collective even at world size 1 (exercises the RCCL path on one GPU)."""
    self.group = process_group
    self.reduce_dtype = reduce_dtype
    self.always_reduce = always_reduce
    # defer=True: mark_ready only counts; the collectives start when the caller says so (launch_all) -- used to keep RCCL's
    # kernels off the chip while the one-workgroup-per-CU GEMMs of the ResnetBlocks run (CU contention)
    self.defer = defer
    items = [(n, p) for n, p in named_params if p.requires_grad]
    if reverse:
      items = items[::-1]
    self.buckets = []        # dict(flat, params[(name,param,offset,numel)], pending, handle)
    cur, cur_bytes = [], 0
    for n, p in items:
      nb = p.numel() * p.element_size()
      if cur and cur_bytes + nb > bucket_bytes:
        self.buckets.append(self._make_bucket(cur))
        cur, cur_bytes = [], 0
      cur.append((n, p))
      cur_bytes += nb
    if cur:
      self.buckets.append(self._make_bucket(cur))
    self._where = {}
    for bi, b in enumerate(self.buckets):
      for (n, p, off, numel) in b['params']:
        self._where[id(p)] = bi
    self.reset()

  @staticmethod
  def _make_bucket(items):
    dev, dt = items[0][1].device, items[0][1].dtype    # fp32 masters on the GPU (fp64 in CPU tests)
    total = sum(((p.numel() + 3) // 4) * 4 for _, p in items)   # keep every view >= 16-byte aligned
    total = (total + 7) // 8 * 8                                # whole 16-byte bf16 vectors for the wire copy
    flat = torch.zeros(total, dtype=dt, device=dev)
    params, off = [], 0
    for n, p in items:
      numel = p.numel()
      view = flat.as_strided(p.shape, p.stride(), off)   # same memory order as the parameter
      if p.grad is not None:
        view.copy_(p.grad)
      p.grad = view
      params.append((n, p, off, numel))
      off += ((numel + 3) // 4) * 4
    return dict(flat=flat, params=params, pending=len(params), handle=None, wire=None)

  def reset(self):
    for b in self.buckets:
      b['pending'] = len(b['params'])
      b['handle'] = None

  # What the last finish() saw (tests/test_hip_ddp.py asserts on it; bench.py prints it for N > 1): how many collectives were
  # started, how many of them had been started before finish() itself had to start them, how many handles were waited for.
  stats = None

  def world_size(self):
    return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

  def mark_ready(self, param):
    """Called right after `param.grad` has been (over)written by a backward kernel."""
    bi = self._where.get(id(param))
    if bi is None:
      return
    b = self.buckets[bi]
    b['pending'] -= 1
    if b['pending'] == 0 and not self.defer:
      self._launch(b)

  def launch_all(self):
    """Start the all-reduce of every bucket that has not been started, in bucket order (asynchronous; `finish` waits)."""
    for b in self.buckets:
      if b['handle'] is None:
        self._launch(b)

  def _active(self):
    return dist.is_available() and dist.is_initialized() and (self.world_size() > 1 or self.always_reduce)

  @staticmethod
  def _convert(src, dst):
    if src.is_cuda:
      from . import ops
      ops.cast_(src, dst)          # HIP kernel on the current stream (the one the gradients were written on)
    else:
      dst.copy_(src)               # CPU rehearsal (gloo tests)

  def _launch(self, b):
    if not self._active() or b['handle'] is not None:
      return
    buf = b['flat']
    if self.reduce_dtype is not None and self.reduce_dtype != buf.dtype:
      if b['wire'] is None:
        b['wire'] = torch.empty(buf.numel(), dtype=self.reduce_dtype, device=buf.device)
      self._convert(buf, b['wire'])
      buf = b['wire']
    b['handle'] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

  def finish(self):
    """Launch whatever has not been launched (parameters without a gradient this step) and make
    the current stream wait for every bucket; no host synchronisation with RCCL."""
    early = sum(1 for b in self.buckets if b['handle'] is not None)
    self.launch_all()
    waited = 0
    for b in self.buckets:
      if b['handle'] is not None:
        b['handle'].wait()
        waited += 1
        if b['wire'] is not None:
          self._convert(b['wire'], b['flat'])
    self.stats = dict(buckets=len(self.buckets), started_before_finish=early, waited=waited)
    self.reset()

  def total_bytes(self):
    return sum(b['flat'].numel() * b['flat'].element_size() for b in self.buckets)
