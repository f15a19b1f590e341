"""GPU: interop pieces of SURVEY.md 8 f2-f4 that round 1 shipped untested -- optimizer-state and VGG-weight ingestion
in the reference's formats, the tolerant network loader, the niter_fix_global phase and its end, lambda annealing, the
plateau scheduler -- plus the device-side evaluation distortion and the fp32 <-> bf16 wire cast."""
import copy
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import jpdse_hip
from jpdse_hip import ops, F32, BF16
from jpdse_hip.optim import FusedAdam
from ctu.trainers import get_trainer
from ctu.models.pix2pixHD_networks import networks
from oracle.ctu_cpu import model as omodel
from oracle.ctu_cpu import nets as onets
from hip_util import DEV, assert_close, to_act, to_nchw, rel_err


def _opts(**kw):
  return omodel.default_opt(gpu_ids=[0], print_losses=False, **kw)


def _paired(kw, seed=1234):
  opt = _opts(**kw)
  torch.manual_seed(seed)
  ora = omodel.OracleTrainer(omodel.default_opt(**kw))
  tr = get_trainer(opt)(opt, 'train')
  tr.model.netG.load_state_dict({k: v.detach() for k, v in ora.G.items()})
  tr.model.netD.load_state_dict({k: v.detach() for k, v in ora.D.items()})
  return tr, ora, opt


# ---- optimizer state in torch.optim.Adam's format (pix2pixHD_trainer.py:124-136,147-148) ---------------------------
def test_fused_adam_resumes_from_torch_adam_checkpoint_and_steps():
  """A reference stats_and_optim.pt holds torch.optim.Adam state of NCHW-contiguous parameters (possibly with an int
  step).  FusedAdam must load it into its channels_last masters and continue exactly like torch.optim.Adam would."""
  g = torch.Generator().manual_seed(3)
  shapes = [(16, 8, 3, 3), (16,), (8, 16, 4, 4)]
  ref = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
  o_ref = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
  grads = [[torch.randn(s, generator=g) for s in shapes] for _ in range(3)]
  for p, gr in zip(ref, grads[0]):
    p.grad = gr.clone()
  o_ref.step()
  sd = copy.deepcopy(o_ref.state_dict())          # state_dict() aliases the optimizer's own state
  sd['state'][0]['step'] = 1                      # old-torch checkpoints: python int
  mine = []
  for p in ref:
    q = torch.nn.Parameter(p.detach().clone().to(DEV))
    if q.dim() == 4:
      q.data = q.data.contiguous(memory_format=torch.channels_last)
    mine.append(q)
  o_hip = FusedAdam(mine, lr=1.0, betas=(0.9, 0.9))
  o_hip.load_state_dict(sd)
  for step in (1, 2):
    for p, q, gr in zip(ref, mine, grads[step]):
      p.grad = gr.clone()
      q.grad = torch.empty_like(q, memory_format=torch.preserve_format).copy_(gr.to(DEV))
    o_ref.step()
    o_hip.step()
  torch.cuda.synchronize()
  for p, q in zip(ref, mine):
    assert_close(q.detach().cpu(), p.detach(), 1e-6, 'parameter after resuming from a torch.optim.Adam checkpoint')
  assert float(o_hip.state[mine[0]]['step']) == 3.0


# ---- torchvision VGG19 weights (networks.py:477-492) ----------------------------------------------------------------
def test_vgg19_ingests_torchvision_state_dict():
  """`models.vgg19().state_dict()` keys are features.<i>.weight/bias with i counting convs, ReLUs and pools
  (networks.py:477-492 slices that Sequential).  Load weights in that layout (the oracle's seeded init under ANOTHER
  seed than the built-in one) and compare the five taps with the oracle's VGG on those weights."""
  sd_o = onets.init_vgg19(seed=77)
  tv = onets.vgg_torchvision_keys(sd_o)
  assert 'features.28.weight' in tv and 'features.0.bias' in tv and len(tv) == 26
  vgg = networks.Vgg19(compute_dtype='fp32', device=DEV)
  before = vgg.convs[3].weight.detach().clone()
  vgg.load_torchvision_state_dict(tv)
  assert not torch.equal(before, vgg.convs[3].weight.detach())
  x = torch.rand(1, 3, 32, 64, generator=torch.Generator().manual_seed(1)) - 0.5
  want = onets.vgg19_features(sd_o, x)
  got = vgg(x.to(DEV))
  assert len(got) == 5
  for k, (a, b) in enumerate(zip(got, want)):
    assert_close(a.cpu(), b, 1e-3, 'relu%d_1 with ingested weights' % (k + 1))


def test_vgg_loss_refuses_silent_random_weights():
  """Training with the VGG loss and neither --vgg19_state_dict nor --vgg_random_init must not silently optimise
  against random features; --no_vgg_loss needs neither."""
  with pytest.raises(ValueError, match='vgg19_state_dict'):
    get_trainer(_opts(ngf=8, ndf=8, n_blocks_global=1, vgg_random_init=False))(
        _opts(ngf=8, ndf=8, n_blocks_global=1, vgg_random_init=False), 'train')
  o = _opts(ngf=8, ndf=8, n_blocks_global=1, vgg_random_init=False, no_vgg_loss=True)
  get_trainer(o)(o, 'train')


def test_vgg19_state_dict_flag_loads_file(tmp_path):
  tv = dict(onets.vgg_torchvision_keys(onets.init_vgg19(seed=5)))
  path = os.path.join(str(tmp_path), 'vgg19.pth')
  torch.save(tv, path)
  o = _opts(ngf=8, ndf=8, n_blocks_global=1, vgg_random_init=False, vgg19_state_dict=path)
  tr = get_trainer(o)(o, 'train')
  assert_close(tr.model.criterionVGG.vgg.convs[0].weight.detach().cpu(), tv['features.0.weight'], 1e-7, 'conv1_1')


# ---- tolerant loader (base_model.py:62-97) --------------------------------------------------------------------------
def test_tolerant_loader_subset_superset_and_shape_mismatch(tmp_path, capsys):
  kw = dict(ngf=8, ndf=8, n_blocks_global=2)
  tr, ora, opt = _paired(kw)
  tr.opt.save_dir = str(tmp_path)
  tr.save(0, 1.0)
  full = torch.load(os.path.join(str(tmp_path), 'net_G.pth'))
  own = {k: v.clone() for k, v in tr.model.netG.state_dict().items()}

  def reload(sd, **over):
    torch.save(sd, os.path.join(str(tmp_path), 'net_G.pth'))
    o = _opts(load_model=True, checkpoints_dir=str(tmp_path), **dict(kw, **over))
    torch.manual_seed(99)                                   # fresh weights differ from the checkpoint
    return get_trainer(o)(o, 'train').model.netG.state_dict()

  # (1) checkpoint with EXTRA layers (a 3-block generator's file into a 2-block network cannot match by key; emulate
  #     the reference case: extra keys that the network does not own are ignored)
  extra = dict(full)
  extra['model.99.weight'] = torch.zeros(3, 3, 1, 1)
  got = reload(extra)
  assert all(torch.equal(got[k].cpu(), full[k]) for k in full)
  assert 'excessive layers' in capsys.readouterr().out
  # (2) checkpoint with FEWER layers: what is there loads, the rest keeps its fresh initialisation and is reported
  fewer = {k: v for k, v in full.items() if not k.startswith('model.1.')}
  got = reload(fewer)
  assert all(torch.equal(got[k].cpu(), full[k]) for k in fewer)
  assert not torch.equal(got['model.1.weight'].cpu(), full['model.1.weight'])
  out = capsys.readouterr().out
  assert 'fewer layers' in out and 'model' in out
  # (3) a tensor whose SHAPE differs (other input channel count) is skipped, not crashed on
  other = dict(full)
  other['model.1.weight'] = torch.zeros(8, 12, 7, 7)
  got = reload(other)
  assert got['model.1.weight'].shape == own['model.1.weight'].shape
  assert torch.equal(got['model.4.weight'].cpu(), full['model.4.weight'])
  # (4) a missing generator checkpoint is an error, a missing discriminator one is not
  os.remove(os.path.join(str(tmp_path), 'net_G.pth'))
  with pytest.raises(FileNotFoundError):
    o = _opts(load_model=True, checkpoints_dir=str(tmp_path), **kw)
    get_trainer(o)(o, 'train')


# ---- coarse-to-fine schedule (pix2pixHD_model.py:248-268,795-804) ---------------------------------------------------
def test_niter_fix_global_trains_only_the_enhancer_then_everything():
  kw = dict(netG='local', ngf=4, ndf=4, n_downsample_global=2, n_blocks_global=1, n_blocks_local=1, niter_fix_global=3)
  tr, ora, opt = _paired(kw)
  trained = {id(p) for grp in tr.optimizer_G.param_groups for p in grp['params']}
  names = {k for k, p in tr.model.netG.named_parameters() if id(p) in trained}
  assert names and all(k.startswith('model1') for k in names)
  assert {k for k in ora.G if k.startswith('model1')} == names
  before = {k: v.detach().clone() for k, v in tr.model.netG.state_dict().items()}
  xd = omodel.synthetic_batch(2, 32, 64, seed=3)
  tr.step(xd)
  ora.step(xd)
  after = tr.model.netG.state_dict()
  for k in before:
    if k.startswith('model1'):
      if k.endswith('.weight'):
        assert not torch.equal(before[k], after[k]), k
        a, b = after[k].cpu().double(), ora.G[k].detach().double()
        assert ((a - b).norm() / b.norm()).item() <= 3e-3, k
    else:
      assert torch.equal(before[k], after[k]), 'fixed coarse-generator tensor %s moved' % k
  for k in omodel.LOSS_NAMES:
    assert abs(tr.last_losses[k] - ora.last_losses[k]) <= 1e-3 * max(abs(ora.last_losses[k]), 1e-6), k
  # end of the phase: all generator parameters train (fresh Adam, beta2 = 0.999 as in the reference)
  new_opt = tr.update_fixed_params()
  assert new_opt is tr.optimizer_G
  assert {id(p) for grp in new_opt.param_groups for p in grp['params']} == {id(p) for p in tr.model.netG.parameters()}
  ora_all = torch.optim.Adam(list(ora.G.values()), lr=opt.lr, betas=(opt.beta1, 0.999))
  ora.optimizer_G = ora_all
  for k, v in tr.model.netG.state_dict().items():          # same starting point for the joint step
    ora.G[k].data.copy_(v.cpu())
  for k, v in tr.model.netD.state_dict().items():
    ora.D[k].data.copy_(v.cpu())
  mid = {k: v.detach().clone() for k, v in tr.model.netG.state_dict().items()}
  xd2 = omodel.synthetic_batch(2, 32, 64, seed=4)
  tr.step(xd2)
  ora.step(xd2)
  after = tr.model.netG.state_dict()
  for k in mid:
    if k.endswith('.weight'):
      assert not torch.equal(mid[k], after[k]), 'after update_fixed_params %s must train' % k
      a, b = after[k].cpu().double(), ora.G[k].detach().double()
      assert ((a - b).norm() / b.norm()).item() <= 3e-3, k


def test_lambda_annealing_and_plateau_scheduler(tmp_path):
  """--anneal_lambda multiplies the distortion weight every anneal_interval steps (pix2pixHD_trainer.py:81-82) and is
  carried by checkpoints; --schedule_lr drives two ReduceLROnPlateau schedulers (pix2pixHD_trainer.py:21-26,28-30)."""
  kw = dict(ngf=8, ndf=8, n_blocks_global=1, anneal_lambda=True, anneal_interval=2, anneal_factor=3.0,
            schedule_lr=True, lr_decay_factor=0.5, lr_decay_patience=0,
            # distortion only: the generator gradient is then exactly lambda_distortion * weight * dL1, so the annealed
            # weight is visible in the gradient norm (an L1 gradient has a sign pattern, i.e. a near-constant norm)
            no_g_gan_loss=True, no_vgg_loss=True, no_gan_feat_loss=True)
  tr, ora, opt = _paired(kw)
  xd = omodel.synthetic_batch(1, 32, 64, seed=2)
  weights, gnorm = [], []
  head = dict(tr.model.netG.named_parameters())['model.10.weight']
  for s in range(4):
    weights.append((tr.lambda_distortion_weight, ora.lambda_distortion_weight))
    tr.model.netG.load_state_dict({k: v.detach() for k, v in ora.G.items()})     # same weights before every step
    tr.step(xd)
    ora.step(xd, keep_grads=True)
    torch.cuda.synchronize()
    gnorm.append(float(head.grad.double().norm()))
    ref = float(ora.grads_G['model.10.weight'].double().norm())
    assert abs(gnorm[-1] - ref) <= 2e-2 * ref, (s, gnorm[-1], ref)
  assert [w[0] for w in weights] == [1.0, 1.0, 3.0, 3.0] and all(a == b for a, b in weights)
  # (each step's gradient norm matched the oracle's, whose objective carries weight 1, 1, 3, 3 by construction)
  assert tr.lambda_distortion_weight == 9.0
  # plateau scheduler: a worse validation loss halves both learning rates (patience 0)
  lr0 = tr.optimizer_G.param_groups[0]['lr']
  tr.scheduler_step(1.0)
  tr.scheduler_step(2.0)
  assert tr.optimizer_G.param_groups[0]['lr'] == pytest.approx(lr0 * 0.5)
  assert tr.optimizer_D.param_groups[0]['lr'] == pytest.approx(lr0 * 0.5)
  # the fused kernel uses the scheduled lr: Adam's update is linear in lr, so the same step taken at lr0 / 2 and -- from the
  # same weights, gradients and optimizer state -- at lr0 moves every weight exactly half as far
  import copy
  w = dict(tr.model.netG.named_parameters())['model.1.weight']
  st = tr.optimizer_G.state[w]
  tr.step(xd)                                              # writes the gradients the two updates below use
  w0 = w.detach().clone()
  saved = {k: (v.clone() if torch.is_tensor(v) else copy.deepcopy(v)) for k, v in st.items()}
  steps = {q: tr.optimizer_G.state[q]['step'].clone() for q in tr.optimizer_G.state}     # one shared step count
  tr.optimizer_G.step()
  torch.cuda.synchronize()
  half = (w.detach() - w0).clone()
  with torch.no_grad():
    w.copy_(w0)
  for k, v in saved.items():
    st[k] = v.clone() if torch.is_tensor(v) else v
  for q, v in steps.items():
    tr.optimizer_G.state[q]['step'] = v.clone()
  tr.optimizer_G.param_groups[0]['lr'] = lr0
  tr.optimizer_G.step()
  torch.cuda.synchronize()
  full = w.detach() - w0
  tr.optimizer_G.param_groups[0]['lr'] = lr0 * 0.5
  assert half.abs().max().item() > 0
  assert ((half - 0.5 * full).abs().max() / full.abs().max()).item() < 1e-3
  # both survive a checkpoint round trip
  tr.opt.save_dir = str(tmp_path)
  tr.save(3, 0.5)
  o2 = _opts(load_model=True, checkpoints_dir=str(tmp_path), **kw)
  tr2 = get_trainer(o2)(o2, 'train')
  tr2.load()
  assert tr2.lambda_distortion_weight == tr.lambda_distortion_weight
  assert tr2.optimizer_G.param_groups[0]['lr'] == pytest.approx(lr0 * 0.5)
  assert tr2.scheduler_G.state_dict()['best'] == tr.scheduler_G.state_dict()['best']
  assert tr2.start_epoch == 4 and tr2.steps_taken == 5 and tr2.lambda_distortion_weight == 9.0


# ---- evaluation distortion on the device (misc.py:64-95, pix2pixHD_model.py:636-641) --------------------------------
@pytest.mark.parametrize('dtype', [F32, BF16])
@pytest.mark.parametrize('mse', [False, True])
def test_quant_loss_is_integer_exact(dtype, mse):
  g = torch.Generator().manual_seed(5)
  a = (torch.rand(2, 3, 37, 53, generator=g) - 0.5) * 1.3          # beyond [-0.5, 0.5]: exercises the clip
  b = torch.rand(2, 3, 37, 53, generator=g) - 0.5
  a.view(-1)[:5] = torch.tensor([-0.5, 0.5, 0.0, 127.0 / 255 - 0.5, 0.49999997])    # exact quantisation boundaries
  opt = omodel.default_opt(normalize_mean=[0.5, 0.45, 0.55], normalize_std=[1.0, 0.9, 1.1])
  A, B = to_act(a, dtype), to_act(b, F32)
  a_seen = to_nchw(A)                                                # what the device holds (bf16-rounded in bf16 mode)
  qa = omodel.tensor2im(a_seen, opt).astype(np.int64)
  qb = omodel.tensor2im(b, opt).astype(np.int64)
  d = qa - qb
  want = float((d * d).sum() if mse else np.abs(d).sum()) / d.size
  slot = torch.zeros(1, dtype=torch.float32, device=DEV)
  ops.quant_loss(A, B, opt.normalize_mean, opt.normalize_std, mse, slot)
  got = float(slot.item())
  assert got == np.float32(want), (got, want)


def test_cast_roundtrip():
  x = torch.randn(4096 + 8, device=DEV)
  y = torch.empty(x.numel(), dtype=torch.bfloat16, device=DEV)
  ops.cast_(x, y)
  assert torch.equal(y, x.to(torch.bfloat16))
  z = torch.empty_like(x)
  ops.cast_(y, z)
  assert torch.equal(z, x.to(torch.bfloat16).float())
