"""ctypes binding of libjpdse_hip.so (C ABI: include/jpdse.h).

The library is the product; there is deliberately NO fallback.  If the shared object is
missing or a call fails, this module raises -- it never routes to torch ops or to oracle/.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# JPDSE_HIP_LIB: developer override (same-box A/B of two builds of this library); the default is the in-tree build
LIB_PATH = os.environ.get('JPDSE_HIP_LIB') or os.path.join(_HERE, 'libjpdse_hip.so')

F32, BF16 = 0, 1
PAD_ZERO, PAD_REFLECT = 0, 1
ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH = 0, 1, 2, 3


class JpdseError(RuntimeError):
  pass


class ConvDesc(ctypes.Structure):
  _fields_ = [('dtype', ctypes.c_int32), ('N', ctypes.c_int32), ('H', ctypes.c_int32),
              ('W', ctypes.c_int32), ('C', ctypes.c_int32), ('K', ctypes.c_int32),
              ('R', ctypes.c_int32), ('S', ctypes.c_int32), ('stride', ctypes.c_int32),
              ('pad', ctypes.c_int32), ('pad_mode', ctypes.c_int32), ('act', ctypes.c_int32),
              ('slope', ctypes.c_float)]


class InormDesc(ctypes.Structure):
  _fields_ = [('dtype', ctypes.c_int32), ('N', ctypes.c_int32), ('H', ctypes.c_int32),
              ('W', ctypes.c_int32), ('C', ctypes.c_int32), ('act', ctypes.c_int32),
              ('slope', ctypes.c_float), ('eps', ctypes.c_float), ('has_residual', ctypes.c_int32)]


class PackEntry(ctypes.Structure):
  _fields_ = [('w', ctypes.c_void_p), ('out', ctypes.c_void_p)] + \
             [(n, ctypes.c_int32) for n in ('K', 'Ks', 'C', 'Cs', 'R', 'S', 'st', 'qh', 'qw', 'Uh', 'Uw', 'Lk', 'gx', 'gy',
                                            'out_f32', 'reserved_')] + \
             [('blocks', ctypes.c_int64), ('block0', ctypes.c_int64)]


class LossTerm(ctypes.Structure):
  _fields_ = [('partial', ctypes.c_void_p), ('n', ctypes.c_int32), ('inv_count', ctypes.c_float), ('out', ctypes.c_void_p)]


class AdamEntry(ctypes.Structure):
  _fields_ = [('p', ctypes.c_void_p), ('g', ctypes.c_void_p), ('m', ctypes.c_void_p),
              ('v', ctypes.c_void_p), ('n', ctypes.c_int64), ('block0', ctypes.c_int64),
              ('cast_bf16', ctypes.c_void_p)]


_P, _I32, _I64, _F, _SZ = (ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float,
                           ctypes.c_size_t)
_CD, _ND = ctypes.POINTER(ConvDesc), ctypes.POINTER(InormDesc)

# name -> (restype, argtypes); every symbol declared in include/jpdse.h
SIGNATURES = {
    'jpdse_version': (_I32, []),
    'jpdse_last_error': (ctypes.c_char_p, []),
    'jpdse_arch_check': (_I32, [_I32]),
    'jpdse_prof_select': (_I32, [_I32, _I32, _I64, _I32]),
    'jpdse_prof_collect': (_I32, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I64)]),
    'jpdse_prof_collect_class': (_I32, [_I32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I64)]),
    'jpdse_prof_hbm_select': (_I32, [_I32, _I32]),
    'jpdse_prof_hbm_collect': (_I32, [_I32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_I64)]),
    'jpdse_conv_out_shape': (_I32, [_CD, ctypes.POINTER(_I32), ctypes.POINTER(_I32)]),
    'jpdse_conv_plan_query': (_I32, [_CD, ctypes.POINTER(_I32), _I32]),
    'jpdse_conv_fwd_pack_size': (_SZ, [_CD]),
    'jpdse_conv_dgrad_pack_size': (_SZ, [_CD]),
    'jpdse_conv_pack_entries': (_I32, [_CD, _P, _P, ctypes.POINTER(PackEntry), _I32]),
    'jpdse_conv_pack_run': (_I32, [_P, _I32, _I64, _P]),
    'jpdse_conv_pack_weights': (_I32, [_CD, _P, _P, _P, _P]),
    'jpdse_conv_workspace_size': (_SZ, [_CD]),
    'jpdse_conv_fwd': (_I32, [_CD, _P, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_conv_fwd_pool': (_I32, [_CD, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_conv_dgrad': (_I32, [_CD, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_conv_dgrad_relu': (_I32, [_CD, _P, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_conv_dgrad_fused': (_I32, [_CD, _P, _P, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_conv_dgrad_fused_lrelu': (_I32, [_CD, _P, _P, _P, _F, _P, _P, _P, _SZ, _P]),
    'jpdse_conv_dgrad_nsum_slots': (_I32, [_CD]),
    'jpdse_conv_dgrad_fused_nsums': (_I32, [_CD, _P, _P, _P, _P, _P, _P, _P, _I32, _F, _P, _P, _SZ, _P]),
    'jpdse_conv_wgrad': (_I32, [_CD, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_convT_fwd': (_I32, [_CD, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_conv_moment_slots': (_I32, [_CD]),
    'jpdse_convT_moment_slots': (_I32, [_CD]),
    'jpdse_conv_fwd_moments': (_I32, [_CD, _P, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_convT_fwd_moments': (_I32, [_CD, _P, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_inorm_fwd_from_moments': (_I32, [_ND, _P, _P, _I32, _P, _P, _P, _P]),
    'jpdse_convT_dgrad': (_I32, [_CD, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_convT_wgrad': (_I32, [_CD, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_inorm_workspace_size': (_SZ, [_ND]),
    'jpdse_inorm_fwd': (_I32, [_ND, _P, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_inorm_bwd': (_I32, [_ND, _P, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_inorm_bwd_from_sums': (_I32, [_ND, _P, _P, _P, _P, _I32, _P, _P, _SZ, _P]),
    'jpdse_avgpool3s2_fwd': (_I32, [_I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    'jpdse_avgpool3s2_bwd': (_I32, [_I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    'jpdse_maxpool2_fwd': (_I32, [_I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    'jpdse_maxpool2_bwd': (_I32, [_I32, _I32, _I32, _I32, _I32, _P, _P, _P, _P]),
    'jpdse_act_bwd': (_I32, [_I32, _I64, _I32, _F, _P, _P, _P, _P]),
    'jpdse_add': (_I32, [_I32, _I64, _P, _P, _P, _P]),
    'jpdse_channel_sum_workspace_size': (_SZ, [_I64, _I32]),
    'jpdse_channel_sum': (_I32, [_I32, _I64, _I32, _P, _P, _P, _SZ, _P]),
    'jpdse_channel_copy': (_I32, [_I32, _I64, _P, _I32, _I32, _P, _I32, _I32, _I32, _P]),
    'jpdse_concat_channels': (_I32, [_I32, _I64, _P, _I32, _P, _I32, _I32, _I32, _P, _P]),
    'jpdse_copy': (_I32, [_I64, _P, _P, _P]),
    'jpdse_zero': (_I32, [_I32, _I64, _P, _P]),
    'jpdse_cast': (_I32, [_I32, _I32, _I64, _P, _P, _P]),
    'jpdse_quant_loss_workspace_size': (_SZ, []),
    'jpdse_quant_loss': (_I32, [_I32, _I32, _I64, _I32, _P, _P, ctypes.POINTER(ctypes.c_double),
                                ctypes.POINTER(ctypes.c_double), _I32, _P, _P, _SZ, _P]),
    'jpdse_nchw_to_nhwc': (_I32, [_I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    'jpdse_nhwc_to_nchw': (_I32, [_I32, _I32, _I32, _I32, _I32, _P, _P, _P]),
    'jpdse_onehot_edge': (_I32, [_I32, _I32, _I32, _I32, _I32, _P, _P, _P, _I32, _P]),
    'jpdse_input_builder': (_I32, [_I32, _I32, _I32, _I32, _I32, _P, _P, _I32, ctypes.POINTER(_P), ctypes.POINTER(_P), _I32, _I32,
                                   _I32, _I32, _P]),
    'jpdse_insert_channels': (_I32, [_I32, _I64, _P, _I32, _P, _I32, _I32, _I32, _P]),
    'jpdse_loss_workspace_size': (_SZ, [_I64]),
    'jpdse_loss_partial_count': (_I32, [_I64]),
    'jpdse_loss_finalize': (_I32, [_P, _I32, _P]),
    'jpdse_l1_fwd': (_I32, [_I32, _I64, _I64, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_l1_bwd': (_I32, [_I32, _I64, _I64, _P, _P, _P, _F, _P, _P]),
    'jpdse_l1_bwd_relu': (_I32, [_I32, _I64, _I64, _P, _P, _P, _F, _P, _P]),
    'jpdse_l1_fwd_bwd': (_I32, [_I32, _I64, _I64, _P, _P, _P, _F, _I32, _P, _P, _SZ, _P]),
    'jpdse_mse_fwd': (_I32, [_I32, _I64, _I64, _P, _P, _P, _P, _SZ, _P]),
    'jpdse_mse_bwd': (_I32, [_I32, _I64, _I64, _P, _P, _P, _F, _P, _P]),
    'jpdse_mse_const_fwd': (_I32, [_I32, _I64, _I32, _F, _P, _P, _P, _SZ, _P]),
    'jpdse_mse_const_bwd': (_I32, [_I32, _I64, _I32, _F, _P, _P, _F, _P, _P]),
    'jpdse_adam_step': (_I32, [_P, _I32, _I64, _F, _F, _F, _F, _I32, _F, _P]),
}

# the developer build (same sources, -DJPDSE_DEV): the shipped ABI plus include/jpdse_dev.h
DEV_LIB_PATH = os.path.join(_HERE, 'libjpdse_hip_dev.so')
DEV_SIGNATURES = {'jpdse_debug_set_fast_path': (_I32, [_I32]),
                  'jpdse_debug_occupy_cus': (_I32, [_I32, _P, _I32, _P])}

_lib = None
_dev = None
_binding_epoch = [0]     # bumped whenever the bound library or its kernel-selection mode changes (dev_mode enter / exit)


def binding_epoch():
  """Changes whenever answers cached from the library (e.g. a layer's moment-slot count) may have become stale."""
  return _binding_epoch[0]


def _load(path, signatures):
  handle = ctypes.CDLL(path)
  for name, (res, args) in signatures.items():
    fn = getattr(handle, name)   # AttributeError here == ABI/header drift: fail loudly
    fn.restype, fn.argtypes = res, args
  return handle


class dev_mode(object):
  """Developer / test tool: run the enclosed calls on libjpdse_hip_dev.so with kernel-selection mode `mode`
  (include/jpdse_dev.h: e.g. 0 = generic kernels only, 6 = no split-K, 19 = 16x16x32 MFMA halo variant).  The shipped
  library has no such switch; on exit the binding is back on it.  Not re-entrant, single threaded."""

  def __init__(self, mode):
    self.mode = int(mode)

  def __enter__(self):
    global _lib, _dev
    import torch  # noqa: F401
    lib()
    if _dev is None:
      if not os.path.isfile(DEV_LIB_PATH):
        raise JpdseError('libjpdse_hip_dev.so not found at %s -- build it with `make -C jpd-se_amd/csrc`' % DEV_LIB_PATH)
      sig = dict(SIGNATURES)
      sig.update(DEV_SIGNATURES)
      _dev = _load(DEV_LIB_PATH, sig)
    self._saved = _lib
    _lib = _dev
    _binding_epoch[0] += 1
    check(_dev.jpdse_debug_set_fast_path(self.mode), 'jpdse_debug_set_fast_path')
    return _dev

  def __exit__(self, *exc):
    global _lib
    _dev.jpdse_debug_set_fast_path(1)
    _lib = self._saved
    _binding_epoch[0] += 1
    return False


def set_dev_mode(mode):
  """Scripts: switch the binding to the developer build for the rest of the process, in kernel-selection mode `mode`."""
  dev_mode(mode).__enter__()


def lib():
  """The loaded library; raises JpdseError when libjpdse_hip.so has not been built."""
  global _lib
  if _lib is None:
    # torch first: it ships its own libamdhip64; loading ours before it would put a second HIP
    # runtime (the system one) into the process, which then sees no device.
    import torch  # noqa: F401
    if not os.path.isfile(LIB_PATH):
      raise JpdseError(
          'libjpdse_hip.so not found at %s -- build it with `python -c "import __graft_entry__ as g; '
          'g.build()"` or `make -C jpd-se_amd/csrc`.  There is no CPU/torch fallback.' % LIB_PATH)
    _lib = _load(LIB_PATH, SIGNATURES)
  return _lib


def last_error():
  return lib().jpdse_last_error().decode('utf-8', 'replace')


def check(rc, what):
  if rc != 0:
    raise JpdseError('%s failed (%d): %s' % (what, rc, last_error()))


def require_gpu(device=0):
  """Fail loudly unless `device` is a gfx950 GPU visible to this process."""
  import torch
  if not torch.cuda.is_available():
    raise JpdseError('no GPU visible: the JPD-SE HIP path has no CPU fallback')
  check(lib().jpdse_arch_check(int(device)), 'jpdse_arch_check')
