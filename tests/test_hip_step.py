"""GPU parity, train-step level: `ctu.trainers.get_trainer(opt)(opt,'train').step()` on the
HIP path against (a) golden step records from the REAL reference and (b) the oracle trainer
stepped side by side; plus get_img / get_eval_loss, checkpoint round trip, bf16 sanity."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import jpdse_hip
from ctu.trainers import get_trainer
from oracle.ctu_cpu import model as omodel
from hip_util import assert_close, rel_err

WEIGHT_TOL = 3e-3     # relative L2 of parameters after Adam steps (fp32); dead biases excluded
GRAD_TOL = 1e-3       # weight gradients, max-abs relative (north_star bound)
LOSS_TOL = 1e-3


def _opts(**kw):
  o = omodel.default_opt(gpu_ids=[0], print_losses=False, **kw)
  return o


def _paired(opt_kw, seed=1234):
  """HIP trainer and oracle trainer holding the same seeded weights."""
  opt = _opts(**opt_kw)
  torch.manual_seed(seed)
  ora = omodel.OracleTrainer(omodel.default_opt(**opt_kw))
  Trainer = get_trainer(opt)
  assert Trainer.__name__ == 'Pix2PixHDTrainer'
  tr = Trainer(opt, 'train')
  tr.model.netG.load_state_dict({k: v.detach() for k, v in ora.G.items()})
  tr.model.netD.load_state_dict({k: v.detach() for k, v in ora.D.items()})
  return tr, ora, opt


def _check_weights(tr, ora, tol, what, steps=1):
  """Post-Adam weights.  Adam normalises every element's update to ~ +-lr, so an element whose
  gradient is pure rounding noise can legitimately land 2*lr away from the oracle (the same
  effect SURVEY.md §7 describes for the dead biases).  Hence: relative L2 error <= tol on every
  weight tensor, and no element further than the 2*lr*steps sign-flip bound."""
  lr = tr.opt.lr
  for net, ref in ((tr.model.netG, ora.G), (tr.model.netD, ora.D)):
    for k, v in net.state_dict().items():
      if not k.endswith('.weight'):
        continue
      a, b = v.cpu().double(), ref[k].detach().double()
      l2 = ((a - b).norm() / b.norm()).item()
      assert l2 <= tol, '%s: %s relative L2 error %.3e' % (what, k, l2)
      assert (a - b).abs().max().item() <= 2.05 * lr * steps, '%s: %s exceeds the Adam sign-flip bound' % (what, k)


def _sync_from_oracle(tr, ora):
  """Copy the oracle's weights and Adam state into the HIP trainer.  Multi-step trajectories of
  two correct implementations separate chaotically (+-lr sign flips, see _check_weights), so
  every step is compared from an identical starting state."""
  tr.model.netG.load_state_dict({k: v.detach() for k, v in ora.G.items()})
  tr.model.netD.load_state_dict({k: v.detach() for k, v in ora.D.items()})
  for opt_h, opt_o, net, ref in ((tr.optimizer_G, ora.optimizer_G, tr.model.netG, ora.G),
                                 (tr.optimizer_D, ora.optimizer_D, tr.model.netD, ora.D)):
    hip_params = dict(net.named_parameters())
    for k, p_ref in ref.items():
      st_o = opt_o.state.get(p_ref)
      if not st_o:
        continue
      st_h = opt_h._ensure_state(hip_params[k])
      st_h['exp_avg'].copy_(st_o['exp_avg'])
      st_h['exp_avg_sq'].copy_(st_o['exp_avg_sq'])
      st_h['step'] = torch.tensor(float(st_o['step']))


def _l2rel(a, b):
  return ((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30)).item()


def _check_grads(tr, ora, xd, what):
  """Weight gradients of the step about to be taken, HIP vs oracle.  Bound: 1e-3 relative
  (north_star) where the problem is well conditioned; the L1 terms (distortion, D-feature and
  VGG matching) have sign() gradients, so on larger tensors two correct fp32 implementations
  differ by ~2*sqrt(fraction of flipped signs).  The yardstick is therefore the fp64 oracle:
  HIP must be as close to it as the fp32 torch-CPU oracle is, within a factor 2 (measured on MI355X,
  gpurun_out of round 2: 0.23 .. 1.40 -- at full width the HIP path, with its two-level fp32 summation, is 4x CLOSER
  to fp64 than torch-CPU fp32, which is why most tensors there miss the direct 1e-3 bound: the oracle is the noisier
  of the two).  The report of which tensors took which route is printed (pytest -rA / -s shows it)."""
  g32, d32 = ora.grads_in_dtype(xd, torch.float32)
  g64, d64 = ora.grads_in_dtype(xd, torch.float64)
  report = []
  for net, r32, r64 in ((tr.model.netG, g32, g64), (tr.model.netD, d32, d64)):
    for k, p in net.named_parameters():
      if not k.endswith('.weight'):
        continue
      direct = rel_err(p.grad.cpu(), r32[k])
      e_hip, e_t32 = _l2rel(p.grad.cpu(), r64[k]), _l2rel(r32[k], r64[k])
      report.append((k, direct, e_hip, e_t32))
      assert direct <= GRAD_TOL or e_hip <= max(GRAD_TOL, 2.0 * e_t32), \
          '%s grad %s: vs oracle-fp32 %.2e; vs fp64 HIP %.2e, torch-fp32 %.2e' % (what, k, direct, e_hip, e_t32)
  # which tensors needed the fp64 yardstick, and by how much (visible with -s / in the failure output)
  escaped = [r for r in report if r[1] > GRAD_TOL]
  import hip_util
  hip_util.record('%s: weight gradients vs fp32 oracle, direct (max over the %d tensors that met it)' % (what, len(report) - len(escaped)),
                  max([r[1] for r in report if r[1] <= GRAD_TOL] or [0.0]), GRAD_TOL)
  hip_util.record('%s: tensors judged against fp64 instead: worst (HIP-vs-fp64) / (torch-fp32-vs-fp64)' % what,
                  max([r[2] / max(r[3], 1e-30) for r in escaped] or [0.0]), 2.0,
                  '%d of %d tensors; worst direct %.2e' % (len(escaped), len(report), max([r[1] for r in escaped] or [0.0])))
  print('%s: %d of %d weight gradients within %.0e of the fp32 oracle directly; %d judged against fp64:'
        % (what, len(report) - len(escaped), len(report), GRAD_TOL, len(escaped)))
  for k, direct, e_hip, e_t32 in escaped:
    print('   %-40s direct %.2e | vs fp64: HIP %.2e, torch-fp32 %.2e (ratio %.2f)' % (k, direct, e_hip, e_t32,
                                                                                     e_hip / max(e_t32, 1e-30)))
  return report


def _golden_steps(golden_dir, name):
  g = np.load(os.path.join(golden_dir, name + '.npz'))
  kw = dict(netG=str(g['opt_netG']), ngf=int(g['opt_ngf']), ndf=int(g['opt_ndf']),
            n_blocks_global=int(g['opt_n_blocks_global']))
  tr, ora, opt = _paired(kw, seed=int(g['seed']))
  assert list(tr.model.netG.state_dict().keys()) == list(g['Gkeys'])
  assert list(tr.model.netD.state_dict().keys()) == list(g['Dkeys'])
  b, h, w = int(g['batch']), int(g['height']), int(g['width'])
  for s in range(int(g['steps'])):
    xd = omodel.synthetic_batch(b, h, w, seed=100 + s, num_labels=opt.num_labels)
    if s > 0:
      _sync_from_oracle(tr, ora)       # the oracle IS the reference's trajectory (bit-exact, see oracle/)
    ret = tr.step(xd)
    if s == 0:     # both sides hold identical weights only before the oracle's first Adam update
      _check_grads(tr, ora, xd, '%s step 0' % name)
    got = [tr.last_losses[k] for k in omodel.LOSS_NAMES]
    # golden (reference) losses: exact at step 0 (same seeded weights on any host); later steps
    # belong to the generating host's trajectory, so they bound the drift only loosely and the
    # strict comparison is with the oracle stepped on this box (below).
    gtol = LOSS_TOL if s == 0 else 3e-2
    np.testing.assert_allclose(got, g['losses:%d' % s], rtol=gtol, err_msg='%s step %d' % (name, s))
    np.testing.assert_allclose(ret, float(g['ret:%d' % s]), rtol=gtol)
    wmask = np.array([k.endswith('.weight') for k in g['Gkeys']])
    norms = np.array([float(v.double().norm()) for v in tr.model.netG.state_dict().values()])
    np.testing.assert_allclose(norms[wmask], g['Gnorm:%d' % s][wmask], rtol=1e-3)
    wmask = np.array([k.endswith('.weight') for k in g['Dkeys']])
    norms = np.array([float(v.double().norm()) for v in tr.model.netD.state_dict().values()])
    np.testing.assert_allclose(norms[wmask], g['Dnorm:%d' % s][wmask], rtol=1e-3)
    ora.step(xd)
    np.testing.assert_allclose(got, [ora.last_losses[k] for k in omodel.LOSS_NAMES], rtol=LOSS_TOL,
                               err_msg='%s step %d vs oracle' % (name, s))
    _check_weights(tr, ora, WEIGHT_TOL, '%s step %d' % (name, s))
  # Inference after the steps.  The golden get_img belongs to the trajectory of the CPU that
  # generated it; on another host the oracle itself drifts from it (Adam sign flips), so the
  # post-training image is compared with the oracle stepped on THIS box from identical weights,
  # and the golden record only loosely (it is pinned exactly by tests/test_oracle_golden.py).
  _sync_from_oracle(tr, ora)
  xd = omodel.synthetic_batch(b, h, w, seed=999, num_labels=opt.num_labels)
  img = tr.get_img(xd)
  assert img.shape == (b, 3, h, w) and img.is_cuda
  assert_close(img.cpu(), ora.get_img(xd), 4e-4, name + ' get_img')     # measured <= 4.1e-5 (profiles/r03_parity_report.txt)
  np.testing.assert_allclose(tr.get_eval_loss(xd), ora.get_eval_loss(xd), rtol=1e-3)
  assert tuple(g['get_img'].shape) == tuple(img.shape)
  return tr


def test_steps_global_generator_golden(golden_dir):
  _golden_steps(golden_dir, 'step_global_ngf8')


def test_steps_local_enhancer_golden(golden_dir):
  _golden_steps(golden_dir, 'step_local_ngf4')


def test_steps_full_width_golden(golden_dir):
  _golden_steps(golden_dir, 'step_global_ngf64_full')


def test_step_vs_oracle_compressed_input_and_mse():
  """use_compressed: the decoded frame feeds G, every loss still compares with the original
  (pix2pixHD_model.py:517-518 vs :711-767); distortion_loss_fn = mse; batch 3, ragged size."""
  kw = dict(ngf=8, ndf=8, n_blocks_global=1, use_compressed=True, distortion_loss_fn='mse')
  tr, ora, opt = _paired(kw)
  for s in range(2):
    xd = omodel.synthetic_batch(3, 48, 80, seed=7 + s)
    if s > 0:
      _sync_from_oracle(tr, ora)
    tr.step(xd)
    ora.step(xd)
    for k in omodel.LOSS_NAMES:
      assert abs(tr.last_losses[k] - ora.last_losses[k]) <= LOSS_TOL * max(abs(ora.last_losses[k]), 1e-6), \
          (s, k, tr.last_losses[k], ora.last_losses[k])
    _check_weights(tr, ora, WEIGHT_TOL, 'compressed/mse step %d' % s)


def test_loss_flags_zero_terms():
  """--no_*_loss flags drop terms from the optimised objective (pix2pixHD_trainer.py:48-56)."""
  kw = dict(ngf=8, ndf=8, n_blocks_global=1, no_g_gan_loss=True, no_d_gan_loss=True, no_vgg_loss=True,
            no_gan_feat_loss=True)
  tr, ora, opt = _paired(kw)
  d_before = {k: v.clone() for k, v in tr.model.netD.state_dict().items()}
  xd = omodel.synthetic_batch(1, 32, 64, seed=3)
  tr.step(xd)
  ora.step(xd)
  _check_weights(tr, ora, WEIGHT_TOL, 'phase-3 flags')
  for k, v in tr.model.netD.state_dict().items():
    assert torch.equal(v, d_before[k]), 'D must not move when no_d_gan_loss is set'


def test_skip_unused_losses_same_update_less_work():
  """Extension flag: with every D / VGG fed loss switched off, not running D and
  VGG at all gives the same G update as the reference schedule; a VGG-only skip keeps the GAN terms."""
  kw = dict(ngf=8, ndf=8, n_blocks_global=1, no_g_gan_loss=True, no_d_gan_loss=True, no_vgg_loss=True,
            no_gan_feat_loss=True, skip_unused_losses=True)
  tr, ora, opt = _paired(kw)
  xd = omodel.synthetic_batch(2, 32, 64, seed=3)
  tr.step(xd)
  ora.step(xd)
  _check_weights(tr, ora, WEIGHT_TOL, 'phase-3 with skipping')
  assert tr.last_losses['G_VGG'] == 0.0 and tr.last_losses['D_real'] == 0.0
  assert abs(tr.last_losses['G_Distortion'] - ora.last_losses['G_Distortion']) <= LOSS_TOL * ora.last_losses['G_Distortion']
  kw = dict(ngf=8, ndf=8, n_blocks_global=1, no_vgg_loss=True, skip_unused_losses=True)   # BASELINE config 2
  tr, ora, opt = _paired(kw)
  tr.step(xd)
  ora.step(xd)
  _check_weights(tr, ora, WEIGHT_TOL, 'no-VGG with skipping')
  for k in omodel.LOSS_NAMES:
    if k != 'G_VGG':
      assert abs(tr.last_losses[k] - ora.last_losses[k]) <= LOSS_TOL * max(abs(ora.last_losses[k]), 1e-6), k


def test_checkpoint_roundtrip_reference_layout(tmp_path):
  kw = dict(ngf=8, ndf=8, n_blocks_global=1)
  tr, ora, opt = _paired(kw)
  xd = omodel.synthetic_batch(1, 32, 64, seed=1)
  tr.step(xd)
  tr.opt.save_dir = str(tmp_path)
  tr.save(0, 1.0)
  sd = torch.load(os.path.join(str(tmp_path), 'net_G.pth'))
  assert list(sd.keys()) == list(ora.G.keys())
  assert all(sd[k].shape == ora.G[k].shape and sd[k].is_contiguous() for k in sd)
  so = torch.load(os.path.join(str(tmp_path), 'stats_and_optim.pt'))
  assert {'epoch', 'steps_taken', 'optimizer_G_state_dict', 'optimizer_D_state_dict', 'best_val_loss'} <= set(so)
  # a second trainer resumes from it and produces the identical next step
  opt2 = _opts(load_model=True, checkpoints_dir=str(tmp_path), **kw)
  tr2 = get_trainer(opt2)(opt2, 'train')
  tr2.load()
  assert tr2.steps_taken == 1 and tr2.start_epoch == 1
  xd2 = omodel.synthetic_batch(1, 32, 64, seed=2)
  a, b = tr.step(xd2), tr2.step(xd2)
  assert abs(a - b) <= 1e-6 * max(abs(a), 1e-6)


def test_bf16_step_tracks_fp32():
  kw = dict(ngf=8, ndf=8, n_blocks_global=2)
  tr32, ora, _ = _paired(kw)
  opt16 = _opts(compute_dtype='bf16', **kw)
  tr16 = get_trainer(opt16)(opt16, 'train')
  tr16.model.netG.load_state_dict(tr32.model.netG.state_dict())
  tr16.model.netD.load_state_dict(tr32.model.netD.state_dict())
  xd = omodel.synthetic_batch(2, 32, 64, seed=11)
  tr32.step(xd)
  tr16.step(xd)
  ora.step(xd)                                   # the CPU oracle on the same weights and batch: the yardstick is not another HIP run
  for k in omodel.LOSS_NAMES:
    a, b, o = tr16.last_losses[k], tr32.last_losses[k], float(ora.last_losses[k])
    assert abs(a - b) <= 5e-2 * max(abs(b), 1e-3), (k, a, b)
    assert abs(a - o) <= 5e-2 * max(abs(o), 1e-3), ('bf16 vs oracle', k, a, o)
    assert abs(b - o) <= 1e-3 * max(abs(o), 1e-3), ('fp32 vs oracle', k, b, o)


def test_bf16_full_width_fast_kernels_in_situ():
  """The production path: ngf=64 generator, bf16, at 128x256 batch 2 -- large enough that the
  LDS-DMA fast kernel, the halo kernel, the stream-K weight gradient and the tap-expanded head
  gradient are the kernels that run (the small tests mostly exercise the generic ones).

  bf16 has no reference counterpart.  Yardsticks, all on the same weights/batch:
    (1) losses within 2 % of the fp32 oracle;
    (2) the fast kernels WITHOUT split-K and without the fused InstanceNorm moments (debug mode 6: same rounding points and
        the same K order as the generic kernels -- on single convs they are bit-identical) against the generic bf16 kernels:
        every weight gradient cosine > 0.995.  With split-K (the default at this size) 0.04 % of a
        conv's outputs move by one bf16 ulp (scripts/diag_splitk.py), and that alone decorrelates the
        deep gradients to cosine ~0.95 between two equally valid bf16 runs (scripts/
        tests/diag/diag_fast_vs_generic.py: both are at 0.897 from fp32): the default path is therefore held to
        "as close to fp32 as the generic kernels, minus 0.02";
    (3) against the fp32 oracle the gradients are only as faithful as bf16 STORAGE of activations and
        activation gradients allows (ReLU-mask / L1-sign flips; measured cosine 0.90-0.92 in the deep
        layers at random init).  The oracle's bf16-storage emulation (oracle.ctu_cpu.nets.storage_bf16)
        reproduces that on the CPU; the HIP path must be at least as close to fp32 as the emulation,
        minus 0.03, and its gradient norms within 5 %."""
  from oracle.ctu_cpu import nets as onets
  tr32, ora, _ = _paired(dict())
  sdG, sdD = tr32.model.netG.state_dict(), tr32.model.netD.state_dict()
  del tr32
  xd = omodel.synthetic_batch(2, 128, 256, seed=21)
  gG, _gD = ora.grads_in_dtype(xd, torch.float32)
  onets.storage_bf16(True)
  try:
    eG, _eD = ora.grads_in_dtype(xd, torch.float32)
  finally:
    onets.storage_bf16(False)
  ora.step(xd)

  def run(fast_mode):
    # mode 1 = the shipped library; the other kernel selections exist only in the developer build (jpdse_dev.h)
    import contextlib
    with (jpdse_hip.dev_mode(fast_mode) if fast_mode != 1 else contextlib.nullcontext()):
      opt16 = _opts(compute_dtype='bf16')
      tr = get_trainer(opt16)(opt16, 'train')
      tr.model.netG.load_state_dict(sdG)
      tr.model.netD.load_state_dict(sdD)
      tr.step(xd)
      torch.cuda.synchronize()
      grads = {k: p.grad.detach().cpu().double().flatten() for k, p in tr.model.netG.named_parameters()
               if k.endswith('.weight')}
      return grads, dict(tr.last_losses)

  cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30))
  g_fast, L = run(1)
  g_nosplit, _ = run(6)
  g_gen, _ = run(0)
  for k in omodel.LOSS_NAMES:
    assert abs(L[k] - ora.last_losses[k]) <= 2e-2 * max(abs(ora.last_losses[k]), 1e-3), (k, L[k], ora.last_losses[k])
  for k, a in g_fast.items():
    ref, emu = gG[k].double().flatten(), eG[k].double().flatten()
    c_kernels = cos(g_nosplit[k], g_gen[k])
    c_ref, c_emu, c_gen = cos(a, ref), cos(emu, ref), cos(g_gen[k], ref)
    assert c_kernels > 0.995, '%s: fast (no split-K) vs generic bf16 kernels cosine %.5f' % (k, c_kernels)
    assert c_ref >= c_gen - 0.02, '%s: default path vs fp32 %.4f, generic kernels vs fp32 %.4f' % (k, c_ref, c_gen)
    assert c_ref >= c_emu - 0.03, '%s: HIP bf16 vs fp32 %.4f, bf16-storage emulation vs fp32 %.4f' % (k, c_ref, c_emu)
    assert abs(float(a.norm() / ref.norm()) - 1.0) < 5e-2, k


def test_bf16_local_enhancer_full_width_in_situ():
  """LocalEnhancer at its production widths (ngf=32: 64-channel ResnetBlocks at half resolution, 32->3 head,
  1024-channel trunk at 1/32 resolution) in bf16 at 128x256: the all-taps weight gradient, the head kernels with
  32-channel inputs and split-K run in situ.  Yardstick: the fp32 HIP path on the same weights and batch
  (itself checked against the oracle by the golden tests): losses within 2 %, every weight gradient with
  cosine >= 0.85 (bf16 storage noise, see test_bf16_full_width_fast_kernels_in_situ) and norm within 6 %."""
  kw = dict(netG='local', ngf=32)
  xd = omodel.synthetic_batch(1, 128, 256, seed=33)
  tr32, ora, opt32 = _paired(kw, seed=77)          # HIP fp32 and the CPU oracle on the same seeded weights
  sdG = {k: v.detach().clone() for k, v in tr32.model.netG.state_dict().items()}     # state_dict() aliases the parameters
  sdD = {k: v.detach().clone() for k, v in tr32.model.netD.state_dict().items()}
  oG, _oD = ora.grads_in_dtype(xd, torch.float32)
  ora.step(xd)
  tr32.step(xd)
  g32 = {k: p.grad.detach().cpu().double().flatten() for k, p in tr32.model.netG.named_parameters() if k.endswith('.weight')}
  L32 = dict(tr32.last_losses)
  del tr32
  # the fp32 HIP run is the yardstick for bf16 below: pin IT to the oracle at this width first (losses 1e-3; gradients by
  # cosine and norm -- two correct fp32 implementations differ in the sign() gradients of the L1 terms, hence not element-wise)
  cos0 = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30))
  for k in omodel.LOSS_NAMES:
    assert abs(L32[k] - ora.last_losses[k]) <= 1e-3 * max(abs(ora.last_losses[k]), 1e-3), ('fp32 vs oracle', k, L32[k], ora.last_losses[k])
  for k, a in g32.items():
    r = oG[k].detach().double().flatten()
    assert cos0(a, r) >= 0.999 and abs(float(a.norm() / r.norm()) - 1.0) < 5e-3, \
        '%s: fp32 HIP vs oracle weight gradient: cosine %.5f, norm ratio %.4f' % (k, cos0(a, r), float(a.norm() / r.norm()))
  opt16 = _opts(compute_dtype='bf16', **kw)
  tr16 = get_trainer(opt16)(opt16, 'train')
  tr16.model.netG.load_state_dict(sdG)
  tr16.model.netD.load_state_dict(sdD)
  tr16.step(xd)
  for k in omodel.LOSS_NAMES:
    assert abs(tr16.last_losses[k] - L32[k]) <= 2e-2 * max(abs(L32[k]), 1e-3), (k, tr16.last_losses[k], L32[k])
    assert abs(tr16.last_losses[k] - ora.last_losses[k]) <= 2e-2 * max(abs(ora.last_losses[k]), 1e-3), ('bf16 vs oracle', k)
  cos = lambda a, b: float(torch.dot(a, b) / (a.norm() * b.norm()).clamp_min(1e-30))
  for k, p in tr16.model.netG.named_parameters():
    if not k.endswith('.weight'):
      continue
    a = p.grad.detach().cpu().double().flatten()
    c = cos(a, g32[k])
    # the deep layers are the lowest: measured per kernel-selection mode (scripts/diag_cos_modes.py, round 3) 0.8505 with the
    # shipped selection, 0.8510 without the row-streaming kernels (mode 29), 0.8531 without the fused InstanceNorm moments
    # (mode 32) -- the noise floor of bf16 storage at random init, not a property of any one kernel.  Round 2's fused moments
    # were taken from the un-rounded accumulators and sat at 0.849 (bound lowered to 0.83 then); with the moments of the
    # stored values (common.h) every mode meets the original 0.85 again.
    assert c >= 0.85, '%s: bf16 vs fp32 weight-gradient cosine %.4f' % (k, c)
    assert abs(float(a.norm() / g32[k].norm()) - 1.0) < 6e-2, k


def test_unsupported_flags_fail_loudly():
  with pytest.raises(NotImplementedError):
    get_trainer(_opts(no_generator_binarization=False))(_opts(no_generator_binarization=False), 'train')
  with pytest.raises(NotImplementedError):
    get_trainer(_opts(pool_size=5))(_opts(pool_size=5), 'train')


@pytest.mark.parametrize('dtype', ['fp32', 'bf16'])
def test_full_size_batch_gradient_is_mean_of_per_image_gradients(dtype):
  """Oracle-free property at BASELINE.json's size (1024x512, full ngf=64 generator, 2-scale D, VGG): no op of the
  loss graph couples samples (InstanceNorm is per (n, c), every loss is a batch mean), so the generator gradient of
  a 2-image batch is the mean of the two single-image gradients -- the premise of sharding images over ranks
  (SURVEY.md 8e).  fp32: all layers to 1e-3.  bf16: the layers next to the output to 2e-2; deeper ones only in
  direction -- 1-ulp differences of bf16-stored gradients (the split-K factor depends on the batch) decorrelate
  chaotically through 40 layers, as between any two correct bf16 runs."""
  opt = _opts(compute_dtype=dtype, use_compressed=True)
  torch.manual_seed(7)
  tr = get_trainer(opt)(opt, 'train')
  m = tr.model
  xd = omodel.synthetic_batch(2, 512, 1024, seed=11)
  w = dict(w_gan=1.0, w_feat=opt.lambda_feat, w_vgg=opt.lambda_feat, w_dist=opt.lambda_distortion)
  top = ['model.38.weight', 'model.38.bias', 'model.34.weight']
  deep = ['model.24.conv_block.5.weight', 'model.16.conv_block.1.weight', 'model.4.weight', 'model.1.weight']
  names = top + deep
  params = dict(m.netG.named_parameters())

  def grads(x_dict):
    state, slots, layout = m._forward_losses(x_dict, grad_w=dict(feat=w['w_feat'], vgg=w['w_vgg'], dist=w['w_dist']))
    assert m.backward_G(state, w['w_gan'], w['w_feat'], w['w_vgg'], w['w_dist'])
    torch.cuda.synchronize()
    return {k: params[k].grad.detach().double().clone() for k in names}

  def one(i):
    return {k: v[i:i + 1] for k, v in xd.items()}

  g_batch = grads(xd)
  g0, g1 = grads(one(0)), grads(one(1))
  for k in names:
    mean = 0.5 * (g0[k] + g1[k])
    err = ((g_batch[k] - mean).norm() / mean.norm().clamp_min(1e-30)).item()
    cos = ((g_batch[k] * mean).sum() / (g_batch[k].norm() * mean.norm()).clamp_min(1e-30)).item()
    print('%s %s: rel L2 %.3e cos %.5f' % (dtype, k, err, cos))
    if dtype == 'fp32':
      assert err <= 1e-3, '%s: batch gradient deviates from the per-image mean by %.3e (relative L2)' % (k, err)
    elif k in top:
      assert err <= 2e-2, '%s: batch gradient deviates from the per-image mean by %.3e (relative L2)' % (k, err)
    else:
      assert cos >= 0.9, '%s: batch gradient points away from the per-image mean (cos %.4f)' % (k, cos)
