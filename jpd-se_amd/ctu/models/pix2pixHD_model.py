"""JPD-SE model on the MI355X-native kernels, behind the reference's `Pix2PixHDModel` API.

Reference: /root/reference/ctu/models/pix2pixHD_model.py
  __init__ :105-220   forward :231-245   create_optimizers :248-280   preprocess :362-412
  discriminate :451-460   _get_img :508-618   get_eval_loss :621-643   get_train_loss :709-771
  get_edges :774-783   save :786-792

What changed structurally (results identical; DESIGN.md "schedule"):
  * no autograd: `train_step` runs an explicit forward/backward program over libjpdse_hip.so;
  * the discriminator sees ONE batched pass [label|fake ; label|real] (InstanceNorm is per
    sample, so this equals the reference's separate passes) and the reference's third pass
    D(label|fake) -- numerically the same forward as D((label|fake).detach()) -- is not
    recomputed: its gradient w.r.t. the fake image is back-propagated through the saved fake
    half with weight gradients skipped (the reference computes and then discards them);
  * the one-hot / edge / concat input builder writes the 39-channel NHWC tensor directly.
Only the flag subset used by scripts/pix2pixHD_bpg_train.sh is accelerated; other branches
raise NotImplementedError.
"""
import os

import torch

from jpdse_hip import F32, BF16, JpdseError, require_gpu
from jpdse_hip import ops
from jpdse_hip.ops import Act
from jpdse_hip.optim import FusedAdam
from jpdse_hip.layers import PackBatcher
from ctu.utils.image_pool import ImagePool
from ctu.models.pix2pixHD_networks.base_model import BaseModel
from ctu.models.pix2pixHD_networks import networks

LOSS_NAMES = ('G_GAN', 'G_GAN_Feat', 'G_VGG', 'G_Distortion', 'D_real', 'D_fake')


class Pix2PixHDModel(BaseModel):

  @staticmethod
  def modify_commandline_options(parser, train):
    """Every flag of the reference setter (pix2pixHD_model.py:21-102) with the same name, type, default and choices
    (pinned by tests/golden/option_setter_flags.json, dumped from the reference), so that a reference command line or
    opt.pkl parses unchanged; flags of branches outside the accelerated path are accepted here and refused in
    __init__ when they would change the computation.  Extensions: --compute_dtype, --skip_unused_losses,
    --vgg19_state_dict, --vgg_random_init."""
    a = parser.add_argument
    a('--num_D', type=int, default=2)
    a('--n_layers_D', type=int, default=3)
    a('--ndf', type=int, default=64)
    a('--no_lsgan', action='store_true')
    a('--pool_size', type=int, default=0)
    a('--no_instance', action='store_true')
    a('--no_label', action='store_true')
    a('--sem_masking', action='store_true')
    a('--norm', type=str, default='instance')
    a('--lambda_feat', type=float, default=10.0)
    a('--lambda_distortion', type=float, default=10.0)
    a('--anneal_lambda', action='store_true')
    a('--anneal_interval', type=int, default=5000)
    a('--anneal_factor', type=float, default=5.)
    a('--match_raw_feat', action='store_true')
    for f in ('gan_feat', 'vgg', 'distortion', 'g_gan', 'd_gan'):
      a('--no_%s_loss' % f, action='store_true')
    a('--data_type', default=32, type=int, choices=[8, 16, 32])
    a('--fp16', action='store_true', default=False)
    a('--skip_unused_losses', action='store_true',
      help='extension (SURVEY 8f-4): do not run D / VGG at all when every loss they feed is switched off by '
           '--no_*_loss (the reference runs them and discards the result, pix2pixHD_trainer.py:48-56); '
           'the skipped losses are reported as 0')
    a('--compute_dtype', type=str, default='fp32', choices=['fp32', 'bf16'],
      help='MFMA input type of the HIP kernels (fp32 accumulate either way)')
    a('--local_rank', type=int, default=0)
    a('--input_nc', type=int, default=3)
    a('--use_compressed', action='store_true')
    a('--ext', type=str, default='jpg', choices=['jpg', 'j2k', 'bpg', 'webp'])
    a('--quality', type=str, default='100')
    a('--zero_sem', action='store_true')
    a('--zero_ins', action='store_true')
    a('--zero_vis', action='store_true')
    a('--checkpoints_dir', type=str)
    a('--netG', type=str, default='global')
    a('--ngf', type=int, default=64)
    a('--n_downsample_global', type=int, default=4)
    a('--n_blocks_global', type=int, default=9)
    a('--n_blocks_local', type=int, default=3)
    a('--n_local_enhancers', type=int, default=1)
    a('--niter_fix_global', type=int, default=0)
    a('--no_feat_encoding', action='store_true')
    a('--no_feat', action='store_true')
    a('--no_label_encoding', action='store_true')
    a('--no_generator_binarization', action='store_true')
    a('--bin_generator_before_res', action='store_true')
    a('--generator_binarizer_out_channels', type=int, default=128)
    a('--use_netE_output', action='store_true')
    # learned-codec / masking branches: parsed for command-line compatibility only
    a('--binary_mask', action='store_true')
    a('--netE_groups', type=int, default=1)
    a('--inst_wise_pool', action='store_true')
    a('--use_dropout', action='store_true',
      help='declared by the reference but never read there (ResnetBlock is always built without dropout)')
    a('--feat_num', type=int, default=3)
    a('--n_downsample_E', type=int, default=4)
    a('--nef', type=int, default=64)
    a('--label_encoder_out_channels', type=int, default=36)
    a('--n_downsample_E4label', type=int, default=4)
    a('--ne4lf', type=int, default=64)
    a('--no_encoder_binarization', action='store_true')
    a('--encoder_binarizer_out_channels', type=int, default=128)
    a('--no_label_encoder_binarization', action='store_true')
    a('--label_encoder_binarizer_out_channels', type=int, default=128)
    a('--vgg19_state_dict', type=str, default=None,
      help='extension: local torchvision-format vgg19 state_dict (keys features.<i>.weight/bias) for the VGG loss; the '
           'reference downloads models.vgg19(pretrained=True) (networks.py:477), which needs network access')
    a('--checkpoint_resblocks', action='store_true',
      help='activation checkpointing of the ResnetBlocks (keep the block input, recompute in backward): the memory saver of '
           'BASELINE config 5; not needed on a 288 GB part, bit-identical results either way')
    a('--ddp_overlap', type=str, default='d_backward', choices=['d_backward', 'layers'],
      help='extension (data parallelism): where the generator gradient all-reduce runs -- behind the discriminator backward '
           '(default; keeps RCCL off the chip while the one-workgroup-per-CU ResnetBlock GEMMs run) or layer by layer from '
           'the backward hooks')
    a('--vgg_random_init', action='store_true',
      help='extension: explicitly accept a seeded random-weight VGG19 for the VGG loss (tests, benchmarks); without '
           'this flag training with the VGG loss and no --vgg19_state_dict is refused')
    return parser

  # ------------------------------------------------------------------------------------------
  def __init__(self, opt):
    super(Pix2PixHDModel, self).__init__(opt)
    g = lambda name, default=False: getattr(opt, name, default)
    unsupported = []
    if not g('no_label_encoding') or not g('no_feat_encoding'):
      unsupported.append('learned label/feature encoders (run with --no_label_encoding --no_feat_encoding)')
    if not g('no_generator_binarization'):
      unsupported.append('generator binarization (run with --no_generator_binarization)')
    for flag in ('sem_masking', 'no_label', 'no_feat', 'match_raw_feat', 'no_lsgan', 'use_netE_output',
                 'zero_sem', 'zero_ins', 'zero_vis', 'binary_mask', 'inst_wise_pool'):
      if g(flag):
        unsupported.append('--' + flag)
    if g('norm', 'instance') != 'instance':
      unsupported.append('--norm ' + str(opt.norm))
    if unsupported:
      raise NotImplementedError('outside the accelerated JPD-SE path (SURVEY.md §2): ' + '; '.join(unsupported))
    self.opt = opt
    self.is_train = opt.is_train
    self.use_features = True
    cd = 'bf16' if (g('compute_dtype', 'fp32') == 'bf16' or g('fp16')) else 'fp32'
    self.cdtype = networks.dtype_code(cd)
    if self.use_gpu():
      require_gpu(opt.gpu_ids[0])

    # channel bookkeeping (pix2pixHD_model.py:117-158)
    self.n_onehot = opt.num_labels + 1 if g('contain_dontcare_label') else opt.num_labels
    self.label_nc = self.n_onehot + (0 if g('no_instance') else 1)
    netG_input_nc = self.label_nc + opt.input_nc
    netD_input_nc = self.label_nc + opt.num_out_channels

    self.netG = networks.define_G(netG_input_nc, opt.num_out_channels, opt.ngf, opt.netG,
                                  opt.n_downsample_global, opt.n_blocks_global, opt.n_local_enhancers,
                                  opt.n_blocks_local, opt.norm, gpu_ids=self.gpu_ids, compute_dtype=cd)
    if g('checkpoint_resblocks', False):
      from jpdse_hip.layers import HipResnetBlock
      for m in self.netG.modules():
        if isinstance(m, HipResnetBlock):
          m.recompute = True
    if self.is_train:
      self.netD = networks.define_D(netD_input_nc, opt.ndf, opt.n_layers_D, opt.norm, False, opt.num_D, True,
                                    gpu_ids=self.gpu_ids, compute_dtype=cd)
    print('---------- networks initialized -------------')
    if not self.is_train or g('load_model'):
      self.load_network(self.netG, 'G', opt)
      if self.is_train:
        self.load_network(self.netD, 'D', opt)
    if self.is_train:
      if opt.pool_size > 0 and len(self.gpu_ids) > 1:
        raise NotImplementedError('Fake Pool Not Implemented for MultiGPU')
      self.fake_pool = ImagePool(opt.pool_size)
      self.criterionGAN = networks.GANLoss(use_lsgan=True)
      self.criterionVGG = networks.VGGLoss(self.gpu_ids, compute_dtype=cd)
      # The reference always optimises against ImageNet VGG19 features (networks.py:477).  A random-weight VGG is a
      # different objective, so it must be asked for by name; it is what tests and bench.py use (same FLOPs / bytes).
      vgg_path = g('vgg19_state_dict', None)
      vgg_needed = not (g('no_vgg_loss') and g('skip_unused_losses'))
      if vgg_path:
        self.criterionVGG.vgg.load_torchvision_state_dict(torch.load(vgg_path, map_location='cpu'))
      elif vgg_needed and not g('no_vgg_loss') and not g('vgg_random_init'):
        raise ValueError('the VGG loss needs ImageNet weights: pass --vgg19_state_dict <torchvision vgg19 state_dict> '
                         '(the reference downloads them, networks.py:477), or --vgg_random_init to accept a seeded '
                         'random-weight VGG19 (tests / benchmarks only), or --no_vgg_loss')
      self.loss_names = LOSS_NAMES
    else:
      self.loss_names = ('G_Distortion')
    if opt.distortion_loss_fn not in ('l1', 'mse'):
      raise ValueError('distortion_loss_fn must be l1 or mse')
    self.grad_buckets = {}           # 'G' / 'D' -> jpdse_hip.ddp.GradBuckets (set by the trainer for DDP)
    self._one = None

  def use_gpu(self):
    return len(self.opt.gpu_ids) > 0

  def _device(self):
    if not self.use_gpu():
      raise JpdseError('the JPD-SE HIP path needs --gpu_ids >= 0: there is no CPU fallback')
    return torch.device('cuda', self.opt.gpu_ids[0])

  # ------------------------------------------------------------------------------------------
  def forward(self, x_dict, opt=None, mode='get_train_loss'):
    if mode == 'get_img':
      return self.get_img(x_dict)
    if mode == 'get_train_loss':
      return self.get_train_loss(x_dict)
    if mode == 'get_eval_loss':
      return self.get_eval_loss(x_dict)
    if mode in ('get_code', 'get_eval_rate'):
      raise NotImplementedError('binary codes exist only for the learned-codec ablations (SURVEY.md §2 row 3)')
    raise ValueError('Invalid forward mode: {}'.format(mode))

  def create_optimizers(self, opt):
    """Two Adams, lr/betas from opt (pix2pixHD_model.py:248-280); with niter_fix_global > 0 only
    the outermost enhancer (`model<n>_*`) is trained."""
    if opt.niter_fix_global > 0:
      prefix = 'model' + str(opt.n_local_enhancers)
      params = [p for k, p in self.netG.named_parameters() if k.startswith(prefix)]
      # the reference leaves requires_grad on and lets autograd compute gradients nobody applies; here the fixed
      # parameters are frozen, so their weight gradients -- and the whole backward of the coarse generator, which
      # only leads to frozen weights -- are not computed at all (LocalEnhancer.bwd)
      keep = {id(p) for p in params}
      for p in self.netG.parameters():
        p.requires_grad_(id(p) in keep)
      print('------------- only training the local enhancer network (for %d epochs) ------------'
            % opt.niter_fix_global)
    else:
      params = list(self.netG.parameters())
    for m in list(self.netG.modules()) + list(self.netD.modules()):
      if hasattr(m, 'ensure_grads'):
        m.ensure_grads()
    optimizer_G = FusedAdam(params, lr=opt.lr, betas=(opt.beta1, opt.beta2))
    optimizer_D = FusedAdam(list(self.netD.parameters()), lr=opt.lr, betas=(opt.beta1, opt.beta2))
    return optimizer_G, optimizer_D

  # ---- external codec hook (config 1 plumbing; CPU, third-party) -----------------------------
  @staticmethod
  def converter(filename, ext, quality):
    """Round-trip `filename` through an outside codec, return the path of the decoded image
    (pix2pixHD_model.py:287-321).  BPG shells out to bpgenc/bpgdec when they are installed."""
    from ctu.utils import codec
    return codec.converter(filename, ext, quality)

  def compress(self, x_dict, tmp_folder=None):
    """Decoded frame(s) of x_dict['image'] after the codec, normalised like the input (pix2pixHD_model.py:324-359).
    Synchronous fallback for batches that arrive without 'compressed_img'; unlike the reference it handles any batch
    size and never shares file names between processes (ctu.utils.codec).  The prefetched form is
    ctu.utils.codec.CodecCollate in the DataLoader."""
    from ctu.utils import codec
    return codec.compress_images(x_dict['image'], self.opt, tmp_folder)

  # ---- input builder ------------------------------------------------------------------------
  def preprocess(self, x_dict, build_base=True):
    """x_dict (CPU or cuda tensors from the loader) -> NHWC device activations.
    Returns dict(base=Act[B,H,W,label_nc+3] with one-hot+edge filled (model.py:375-394) -- None when build_base is False:
    the train step writes the network inputs directly (ops.input_builder) --, real=Act 3ch, src=Act 3ch (decoded frame
    when use_compressed else real), label / inst = the device label and instance maps)."""
    dev = self._device()
    opt = self.opt
    comp = None
    if opt.use_compressed:
      comp = x_dict.get('compressed_img')
      if comp is None:
        comp = self.compress(x_dict, os.path.join(opt.save_dir, 'tmp_imgs'))   # a private sub-directory per process
    label = x_dict['label'].to(dev, dtype=torch.float32, non_blocking=True).contiguous()
    inst = x_dict['instance'].to(dev, dtype=torch.int64, non_blocking=True).contiguous()
    image = x_dict['image'].to(dev, dtype=torch.float32, non_blocking=True).contiguous()
    if getattr(opt, 'no_instance', False):
      inst = torch.zeros_like(inst)      # a constant map has no edges; its channel is not part of label_nc
    total_c = self.label_nc + opt.input_nc
    base = ops.onehot_edge(label, inst, self.n_onehot, total_c, self.cdtype) if build_base else None
    real = ops.nchw_to_nhwc(image, self.cdtype)
    src = real
    if comp is not None:
      src = ops.nchw_to_nhwc(comp.to(dev, dtype=torch.float32, non_blocking=True).contiguous(), self.cdtype)
    return dict(base=base, real=real, src=src, image_nchw=image, label=label, inst=inst, total_c=total_c)

  def _with_image(self, base, img, out=None):
    """torch.cat((input_label, img), dim=1) in NHWC: copy of `base` with the image channels filled."""
    dst = out if out is not None else base.empty_like()
    return ops.concat_channels(base, img, self.label_nc, dst)

  # ---- inference ------------------------------------------------------------------------------
  def get_img(self, x_dict):
    with torch.no_grad():
      pre = self.preprocess(x_dict)
      fake, _ = self.netG.fwd(self._with_image(pre['base'], pre['src']))
      return ops.nhwc_to_nchw(fake)

  def get_eval_loss(self, x_dict):
    """Distortion on de-normalised, clipped, uint8-truncated images (0..255 scale), as the reference computes it on
    the host through tensor2im (pix2pixHD_model.py:636-641, utils/misc.py:64-95) -- here one device pass
    (jpdse_quant_loss: the same float64 arithmetic, bit-identical quantisation, no device->host image copies)."""
    with torch.no_grad():
      pre = self.preprocess(x_dict)
      fake, _ = self.netG.fwd(self._with_image(pre['base'], pre['src']))
      real32 = ops.nchw_to_nhwc(pre['image_nchw'], F32)       # the original image un-rounded, as the reference uses it
      slot = torch.zeros(1, dtype=torch.float32, device=self._device())
      ops.quant_loss(fake, real32, self.opt.normalize_mean, self.opt.normalize_std,
                     self.opt.distortion_loss_fn == 'mse', slot)
      return slot[0]

  # ---- training -------------------------------------------------------------------------------
  def _forward_losses(self, x_dict, grad_w=None):
    """Forward pass of the whole loss graph.  Returns (state for backward, slots tensor, layout).
    grad_w = dict(feat=, vgg=, dist=) (train_step): the gradients of the L1 terms -- whose upstream
    gradient is just that constant weight -- are produced in the same pass as the loss values
    (jpdse_l1_fwd_bwd) and handed to backward_G through `state`."""
    opt = self.opt
    dev = self._device()
    skip = bool(getattr(opt, 'skip_unused_losses', False))
    run_d = not (skip and opt.no_g_gan_loss and opt.no_gan_feat_loss and opt.no_d_gan_loss)
    run_vgg = not (skip and opt.no_vgg_loss)

    # Input builder, one pass (ops.input_builder): the generator input [label | edge | decoded frame] and both halves of the
    # ONE batched discriminator pass [label | edge | fake ; label | edge | real] (model.py:375-394, 595, 456, 717-733) are
    # written straight from the label / instance maps; the fake half gets its image channels once G has produced them.
    pre = self.preprocess(x_dict, build_base=False)
    real, src, label, inst = pre['real'], pre['src'], pre['label'], pre['inst']
    B, H, W = real.N, real.H, real.W
    g_in = Act.empty(B, H, W, pre['total_c'], self.cdtype, dev)
    d_in = Act.empty(2 * B, H, W, pre['total_c'], self.cdtype, dev) if run_d else None
    if run_d:
      ops.input_builder(label, inst, self.n_onehot, [g_in, d_in.batch_slice(B, 2 * B), d_in.batch_slice(0, B)],
                        [src, real, None], self.label_nc)
    else:
      ops.input_builder(label, inst, self.n_onehot, [g_in], [src], self.label_nc)
    fake, g_ctx = self.netG.fwd(g_in)

    pred, d_ctx = None, None
    if run_d:
      ops.insert_channels(d_in.batch_slice(0, B), fake, self.label_nc)
      pred, d_ctx = self.netD.fwd(d_in)

    vf, vr, v_ctx = [], [], None
    n_vgg = len(networks.VGGLoss.weights)
    if run_vgg:
      # VGG19 on the generated and the real image as ONE pass over [fake ; real] (networks.py:124-139 runs vgg(x), vgg(y)):
      # the convolutions are per image, so the two halves of every feature map hold the values of the two separate passes (no
      # kernel on this path lets one image's arithmetic depend on the batch size: equal bit for bit on this build, which
      # tests/test_hip_networks.py::test_vgg19_batched_pass_equals_separate_passes checks -- a property of the kernels chosen,
      # not a contract of the ABI), the launch count halves and the deep, small layers (relu5_1: 32x64 pixels) fill the chip.
      # Backward goes through the fake half of the saved tensors only (Ctx.slice).  Memory: the real half's activations stay
      # alive until the pass ends instead of streaming through -- VGG19 to relu5_1 is ~0.33 GB of bf16 activations per
      # 1024x512 image, so +1.3 GB at batch 4; of no consequence next to 288 GB.
      vgg = self.criterionVGG.vgg
      both = Act.empty(2 * B, H, W, fake.C, self.cdtype, dev)
      ops.copy_(fake.t, both.t[:B])          # two 33 MB device-to-device copies
      ops.copy_(real.t, both.t[B:])
      maps, ctxs = vgg.fwd(both, save=not opt.no_vgg_loss)
      vf = [m.batch_slice(0, B) for m in maps]
      vr = [m.batch_slice(B, 2 * B) for m in maps]
      v_ctx = [c.slice(0, B) if c is not None else None for c in ctxs] if not opt.no_vgg_loss else None

    nD, nF = opt.num_D, opt.n_layers_D + 1
    layout = dict(D_fake=list(range(0, nD)), D_real=list(range(nD, 2 * nD)), G_GAN=list(range(2 * nD, 3 * nD)))
    o = 3 * nD
    layout['feat'] = [[o + i * nF + j for j in range(nF)] for i in range(nD)]
    o += nD * nF
    layout['vgg'] = list(range(o, o + n_vgg))
    o += n_vgg
    layout['dist'] = o
    slots = torch.zeros(o + 1, dtype=torch.float32, device=dev)
    s = lambda i: slots[i:i + 1]
    gw = grad_w or {}
    d_feat, d_vgg, d_dist = None, None, None
    with ops.deferred_loss_finals():          # the second stage of all the loss terms below in ONE launch (jpdse_loss_finalize)
      for i in range(nD if run_d else 0):
        p = pred[i][-1]
        ops.mse_const_fwd(p.batch_slice(0, B), 0.0, s(layout['D_fake'][i]))
        ops.mse_const_fwd(p.batch_slice(B, 2 * B), 1.0, s(layout['D_real'][i]))
        ops.mse_const_fwd(p.batch_slice(0, B), 1.0, s(layout['G_GAN'][i]))
        for j in range(nF):
          f = pred[i][j]
          ff, fr = f.batch_slice(0, B), f.batch_slice(B, 2 * B)
          if gw.get('feat') and j < nF - 1:
            d_feat = d_feat if d_feat is not None else [[None] * nF for _ in range(nD)]
            d_feat[i][j] = ops.l1_fwd_bwd(ff, fr, s(layout['feat'][i][j]), gw['feat'] / nD)
          else:
            ops.l1_fwd(ff, fr, s(layout['feat'][i][j]))
      wk = networks.VGGLoss.weights
      for k in range(len(vf)):
        if gw.get('vgg') and v_ctx is not None:
          d_vgg = d_vgg if d_vgg is not None else [None] * len(vf)
          d_vgg[k] = ops.l1_fwd_bwd(vf[k], vr[k], s(layout['vgg'][k]), gw['vgg'] * wk[k], relu_a=True)
        else:
          ops.l1_fwd(vf[k], vr[k], s(layout['vgg'][k]))
      if opt.distortion_loss_fn == 'l1' and gw.get('dist'):
        d_dist = ops.l1_fwd_bwd(fake, real, s(layout['dist']), gw['dist'])
      else:
        (ops.l1_fwd if opt.distortion_loss_fn == 'l1' else ops.mse_fwd)(fake, real, s(layout['dist']))
    state = dict(B=B, fake=fake, real=real, g_ctx=g_ctx, pred=pred, d_ctx=d_ctx, vf=vf, vr=vr, v_ctx=v_ctx,
                 d_feat=d_feat, d_vgg=d_vgg, d_dist=d_dist)
    return state, slots, layout

  def _reduce_losses(self, vals, layout):
    """Host arithmetic on the slot values -> the reference's six loss scalars."""
    nD = self.opt.num_D
    w = networks.VGGLoss.weights
    return dict(
        G_GAN=sum(vals[i] for i in layout['G_GAN']),
        G_GAN_Feat=sum((1.0 / nD) * vals[i] for row in layout['feat'] for i in row),
        G_VGG=sum(w[k] * vals[i] for k, i in enumerate(layout['vgg'])),
        G_Distortion=vals[layout['dist']],
        D_real=sum(vals[i] for i in layout['D_real']),
        D_fake=sum(vals[i] for i in layout['D_fake']))

  def get_train_loss(self, x_dict):
    """The six losses as 0-dim device tensors in `loss_names` order (values only: the HIP path
    keeps no autograd graph -- use `train_step` to optimise)."""
    state, slots, layout = self._forward_losses(x_dict)
    L = self._reduce_losses(slots.cpu().tolist(), layout)
    return tuple(torch.tensor(L[k], device=slots.device) for k in LOSS_NAMES)

  def _ones(self, dev):
    if self._one is None or self._one.device != dev:
      self._one = torch.ones(1, dtype=torch.float32, device=dev)
    return self._one

  def backward_G(self, state, w_gan, w_feat, w_vgg, w_dist):
    """d(loss_G)/d(netG params), loss_G = w_gan*G_GAN + w_feat*G_GAN_Feat + w_vgg*G_VGG + w_dist*G_Dist."""
    opt, B = self.opt, state['B']
    one = self._ones(state['fake'].t.device)
    fake, real = state['fake'], state['real']
    d_fake = None
    if (w_gan != 0.0 or w_feat != 0.0) and state['pred'] is not None:
      dres = []
      for i in range(opt.num_D):
        row = []
        for j, f in enumerate(state['pred'][i]):
          ff, fr = f.batch_slice(0, B), f.batch_slice(B, 2 * B)
          if j == len(state['pred'][i]) - 1:
            row.append(ops.mse_const_bwd(ff, 1.0, one, w_gan) if w_gan != 0.0 else None)
          elif w_feat == 0.0:
            row.append(None)
          elif state.get('d_feat') is not None and state['d_feat'][i][j] is not None:
            row.append(state['d_feat'][i][j])           # produced with the loss value (l1_fwd_bwd)
          else:
            row.append(ops.l1_bwd(ff, fr, one, w_feat / opt.num_D))
        dres.append(row)
      # only the image channels of D's concatenated input carry a gradient back to G
      d_fake = self.netD.bwd(state['d_ctx'], dres, need_dx=True, need_dw=False, batch=(0, B),
                             dx_channels=(self.label_nc, self.label_nc + fake.C))
    if w_vgg != 0.0 and state['v_ctx'] is not None:
      wk = networks.VGGLoss.weights
      dmaps = state.get('d_vgg') or [ops.l1_bwd(state['vf'][k], state['vr'][k], one, w_vgg * wk[k], relu_a=True)
                                     for k in range(len(wk))]
      dv = self.criterionVGG.vgg.bwd(state['v_ctx'], dmaps)
      d_fake = dv if d_fake is None else ops.add_(d_fake, dv)
    if w_dist != 0.0:
      fn = ops.l1_bwd if opt.distortion_loss_fn == 'l1' else ops.mse_bwd
      dd = state['d_dist'] if state.get('d_dist') is not None else fn(fake, real, one, w_dist)
      d_fake = dd if d_fake is None else ops.add_(d_fake, dd)
    if d_fake is None:
      return False
    self.netG.bwd(state['g_ctx'], d_fake, need_dx=False, need_dw=True)
    return True

  def _repack(self, which):
    """One-launch re-pack of the stepped network's data-gradient panels (jpdse_hip.layers.PackBatcher)."""
    if not hasattr(self, '_pack_batchers'):
      self._pack_batchers = {}
    b = self._pack_batchers.get(which)
    if b is None:
      b = self._pack_batchers[which] = PackBatcher(self.netG if which == 'G' else self.netD)
    b.run()

  def backward_D(self, state, w_d):
    """d(loss_D)/d(netD params), loss_D = w_d * (D_fake + D_real)  (w_d = 0.5 in the trainer)."""
    if w_d == 0.0 or state['pred'] is None:
      return False
    opt, B = self.opt, state['B']
    one = self._ones(state['fake'].t.device)
    dres = []
    for i in range(opt.num_D):
      p = state['pred'][i][-1]
      dp = p.empty_like()
      ops.mse_const_bwd(p.batch_slice(0, B), 0.0, one, w_d, out=dp.batch_slice(0, B))
      ops.mse_const_bwd(p.batch_slice(B, 2 * B), 1.0, one, w_d, out=dp.batch_slice(B, 2 * B))
      dres.append([None] * (len(state['pred'][i]) - 1) + [dp])
    self.netD.bwd(state['d_ctx'], dres, need_dx=False, need_dw=True)
    return True

  def train_step(self, x_dict, optimizer_G, optimizer_D, lambda_distortion_weight=1.0):
    """One optimisation step, results as the reference's (pix2pixHD_trainer.py:44-78: forward, loss_G backward + Adam(G),
    loss_D backward + Adam(D)) -- executed as forward, loss_G backward, loss_D backward, Adam(G), Adam(D): loss_D does not
    depend on G's weights, see below.  One host wait: for the loss copy queued after the forward pass."""
    opt = self.opt
    w_gan = 0.0 if opt.no_g_gan_loss else 1.0
    w_feat = 0.0 if opt.no_gan_feat_loss else opt.lambda_feat
    w_vgg = 0.0 if opt.no_vgg_loss else opt.lambda_feat
    w_dist = 0.0 if opt.no_distortion_loss else opt.lambda_distortion * lambda_distortion_weight
    state, slots, layout = self._forward_losses(x_dict, grad_w=dict(feat=w_feat, vgg=w_vgg, dist=w_dist))
    # All six losses are functions of the forward pass alone (the reference's .item() calls come after the optimizer
    # steps, pix2pixHD_trainer.py:80-85, but read values computed before them), so their device->host copy is queued HERE,
    # behind the loss kernels, and the host waits for THAT copy at the end of the step -- not for the backward passes and
    # Adam updates it has meanwhile enqueued.  step() still returns this step's losses; the host just no longer idles the
    # GPU for a launch round trip between two steps (0.3 ms per step in profiles/r02_step_breakdown.txt).
    early = getattr(self, 'early_loss_readback', True)      # False: copy after the optimizer steps (A/B measurements only)
    if early:
      loss_host, loss_ready = self._queue_loss_readback(slots)
    # Order: both backward passes first, then the two Adams.  The reference steps G before it back-propagates loss_D
    # (pix2pixHD_trainer.py:64-78), but loss_D's graph -- D on fake.detach() and on the real image -- does not contain G's
    # weights, so its gradients are the same numbers either way, bit for bit.  Under data parallelism this gives G's
    # gradient all-reduce (93 % of the bytes) the whole D backward to hide behind (launch_all below), instead of the
    # ResnetBlock GEMMs whose grids are exactly one workgroup per CU and take a second round when RCCL holds any CU.
    bg, bd = self.grad_buckets.get('G'), self.grad_buckets.get('D')
    # Opt-in timeline of the data-parallel schedule (self.ddp_timeline = []: one dict of timing events per step, recorded on the
    # compute stream; bench.py --gpus N > 1 and tests/test_hip_ddp.py read it).  `g_reduced` is recorded right behind the waits
    # on G's all-reduce handles, so elapsed(d_bwd_end -> g_reduced) is what of the collective the D backward did NOT hide.
    tl = getattr(self, 'ddp_timeline', None)
    marks = {} if tl is not None else None

    def mark(name):
      if marks is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        marks[name] = ev

    did_G = self.backward_G(state, w_gan, w_feat, w_vgg, w_dist)
    mark('g_bwd_end')
    if did_G and bg is not None:
      bg.launch_all()                  # no-op for buckets already started from the per-layer hooks (defer=False)
    # (Measured and not adopted, round 4: Adam(G) + the panel re-pack -- 1.2 ms of HBM-bound streaming -- on a second stream
    # beside the discriminator's backward: 26.36-26.40 ms per step against 26.15-26.33 serial, profiles/r04_adam_side_stream_ab.txt.
    # The D backward is itself half HBM-bound passes, and its GEMM launches lose more to the shared CUs than Adam gains.)
    did_D = self.backward_D(state, 0.0 if opt.no_d_gan_loss else 0.5)
    mark('d_bwd_end')
    if did_G:
      if bg is not None:
        bg.finish()
      mark('g_reduced')
      optimizer_G.step()
      self._repack('G')
      mark('adam_g_end')
    if did_D:
      if bd is not None:
        bd.finish()
      mark('d_reduced')
      optimizer_D.step()
      self._repack('D')
    mark('step_end')
    if tl is not None:
      marks['stats_G'] = dict(bg.stats) if bg is not None and bg.stats else None
      marks['stats_D'] = dict(bd.stats) if bd is not None and bd.stats else None
      tl.append(marks)
    if not early:
      loss_host, loss_ready = self._queue_loss_readback(slots)
    loss_ready.synchronize()
    return self._reduce_losses(loss_host.tolist(), layout)

  def _queue_loss_readback(self, slots):
    """Asynchronous copy of the loss slots into pinned host memory on the current stream; returns (host view, event)."""
    n = slots.numel()
    buf = getattr(self, '_loss_pinned', None)
    if buf is None or buf.numel() < n:
      buf = self._loss_pinned = torch.empty(max(n, 64), dtype=torch.float32).pin_memory()
    host = buf[:n]
    host.copy_(slots, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record(torch.cuda.current_stream(slots.device))
    return host, ev

  # ------------------------------------------------------------------------------------------
  def save(self):
    self.save_network(self.netG, 'G', self.opt)
    self.save_network(self.netD, 'D', self.opt)

  def update_fixed_params(self, optimizer_G):
    """After niter_fix_global epochs also fine-tune the coarse generator (model.py:795-804): a fresh Adam over ALL
    generator parameters (beta2 hard-wired to 0.999 as in the reference), carrying the old optimizer's gradient scale.
    Under data parallelism call Pix2PixHDTrainer.update_fixed_params, which also re-buckets the gradients."""
    for p in self.netG.parameters():
      p.requires_grad_(True)
    for m in self.netG.modules():
      if hasattr(m, 'ensure_grads'):
        m.ensure_grads()
    return FusedAdam(list(self.netG.parameters()), lr=self.opt.lr, betas=(self.opt.beta1, 0.999),
                     grad_scale=getattr(optimizer_G, 'grad_scale', 1.0))
