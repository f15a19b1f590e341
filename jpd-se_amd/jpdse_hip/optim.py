"""Fused multi-tensor Adam on the C ABI (`jpdse_adam_step`), a drop-in for the two
`torch.optim.Adam`s of the reference (ctu/models/pix2pixHD_model.py:275,279).

It subclasses torch.optim.Optimizer only for the bookkeeping the reference API exposes:
`param_groups` (ReduceLROnPlateau edits `lr`), `state_dict()` / `load_state_dict()` in
torch.optim.Adam's format (trainer checkpoints: pix2pixHD_trainer.py:124-136).
One kernel launch updates every tensor (HBM-bound: 28 B/param).
"""
import ctypes

import torch

from . import lib, check, AdamEntry
from .layers import bump_weights_epoch


class FusedAdam(torch.optim.Optimizer):

  def __init__(self, params, lr=2e-4, betas=(0.5, 0.999), eps=1e-8, grad_scale=1.0):
    defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                    foreach=None, capturable=False, differentiable=False, fused=None)
    super(FusedAdam, self).__init__(params, defaults)
    self.grad_scale = grad_scale     # e.g. 1/world_size after a SUM all-reduce
    self._table = {}                 # group index -> (ptr signature, device table, n_entries, total_blocks)

  def zero_grad(self, set_to_none=False):
    """Gradients live in persistent buffers that every backward overwrites (beta = 0), so
    there is nothing to clear; kept for API compatibility with the reference's call sites
    (pix2pixHD_trainer.py:64,73)."""
    return None

  def load_state_dict(self, state_dict):
    """torch.optim.Adam checkpoints (the reference's stats_and_optim.pt, pix2pixHD_trainer.py:124-136,147-148) keep
    exp_avg / exp_avg_sq in the memory order the REFERENCE's parameters had (NCHW-contiguous) while the masters here
    are channels_last; torch's loader preserves the saved strides, and the fused kernel needs param / grad / state in
    one memory order.  Re-lay every state tensor into its parameter's strides (values unchanged) and accept the int
    `step` of old-torch checkpoints."""
    super(FusedAdam, self).load_state_dict(state_dict)
    for group in self.param_groups:
      for p in group['params']:
        st = self.state.get(p)
        if not st:
          continue
        for key in ('exp_avg', 'exp_avg_sq'):
          v = st.get(key)
          if v is not None and (v.stride() != p.stride() or v.device != p.device or v.dtype != p.dtype):
            st[key] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(v)
        step = st.get('step', 0)
        st['step'] = torch.tensor(float(step.item() if torch.is_tensor(step) else step), dtype=torch.float32)
    self._table = {}

  def _ensure_state(self, p):
    st = self.state[p]
    if len(st) == 0:
      st['step'] = torch.tensor(0.0, dtype=torch.float32)
      st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
      st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
    return st

  def _build_table(self, gi, group):
    entries, sig, block0 = [], [], 0
    for p in group['params']:
      if p.grad is None:
        continue
      st = self._ensure_state(p)
      g, m, v = p.grad, st['exp_avg'], st['exp_avg_sq']
      if not (p.is_cuda and p.dtype == torch.float32):
        raise RuntimeError('FusedAdam needs fp32 cuda parameters')
      # all four tensors must share one memory order so the update is purely elementwise
      if not (g.stride() == p.stride() and m.stride() == p.stride() and v.stride() == p.stride()):
        raise RuntimeError('FusedAdam: param/grad/state strides differ')
      n = p.numel()
      # conv layers whose forward GEMM panel is a plain bf16 cast of the master (layers.HipConv2d) let the
      # Adam kernel write it: saves the separate cast pass over the weights
      cast = getattr(p, '_jpdse_cast_out', None)
      cast_ptr = cast.data_ptr() if cast is not None else None
      entries.append(AdamEntry(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), n, block0, cast_ptr))
      sig.append((p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), cast_ptr))
      block0 += (n + 1023) // 1024
    return entries, tuple(sig), block0

  @torch.no_grad()
  def step(self, closure=None):
    for gi, group in enumerate(self.param_groups):
      entries, sig, total_blocks = self._build_table(gi, group)
      if not entries:
        continue
      cached = self._table.get(gi)
      if cached is None or cached[0] != sig:
        arr = (AdamEntry * len(entries))(*entries)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        dev = host.to(group['params'][0].device)
        cached = (sig, dev, len(entries), total_blocks)
        self._table[gi] = cached
      _, dev, n_entries, total_blocks = cached
      step_t = None
      for p in group['params']:
        if p.grad is not None:
          st = self.state[p]
          st['step'] += 1
          step_t = st['step']
      b1, b2 = group['betas']
      check(lib().jpdse_adam_step(ctypes.c_void_p(dev.data_ptr()), n_entries, total_blocks, float(group['lr']),
                                  float(b1), float(b2), float(group['eps']), int(step_t.item()),
                                  float(self.grad_scale),
                                  ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), 'adam_step')
    # the kernel updates the masters outside torch's version counter: bump a per-parameter counter
    # (layers.HipConv2d.packs keys its packed GEMM panels on it), only for the tensors that moved
    for group in self.param_groups:
      for p in group['params']:
        if p.grad is not None:
          p._jpdse_wver = getattr(p, '_jpdse_wver', 0) + 1
          if getattr(p, '_jpdse_cast_out', None) is not None:
            p._jpdse_cast_wver = p._jpdse_wver      # forward panel already holds the updated weights
    return None
