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

  def __init__(self, named_params, bucket_bytes=64 << 20, process_group=None, reverse=True):
    """named_params: iterable of (name, param) in FORWARD order."""
    self.group = process_group
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
    return dict(flat=flat, params=params, pending=len(params), handle=None)

  def reset(self):
    for b in self.buckets:
      b['pending'] = len(b['params'])
      b['handle'] = None

  def world_size(self):
    return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

  def mark_ready(self, param):
    """Called right after `param.grad` has been (over)written by a backward kernel."""
    bi = self._where.get(id(param))
    if bi is None:
      return
    b = self.buckets[bi]
    b['pending'] -= 1
    if b['pending'] == 0:
      self._launch(b)

  def _launch(self, b):
    if self.world_size() > 1 and b['handle'] is None:
      b['handle'] = dist.all_reduce(b['flat'], op=dist.ReduceOp.SUM, group=self.group, async_op=True)

  def finish(self):
    """Launch whatever has not been launched (parameters without a gradient this step) and make
    the current stream wait for every bucket; no host synchronisation with RCCL."""
    for b in self.buckets:
      if b['handle'] is None and self.world_size() > 1:
        self._launch(b)
    for b in self.buckets:
      if b['handle'] is not None:
        b['handle'].wait()
    self.reset()

  def total_bytes(self):
    return sum(b['flat'].numel() * b['flat'].element_size() for b in self.buckets)
