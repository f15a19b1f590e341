"""GPU parity, per kernel: every C-ABI op against the torch-CPU oracle primitives on the same
seeded inputs (fp32: <= 1e-3 relative, the north_star bound; bf16: documented looser bound).
Shapes cover each distinct conv class of SURVEY.md §2.1 at sizes the CPU finishes in seconds,
ragged tails (M, K not multiples of the tile), and the borders of pad / pool / convT."""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import jpdse_hip
from jpdse_hip import ops, F32, BF16, PAD_ZERO, PAD_REFLECT, ACT_NONE, ACT_RELU, ACT_LRELU, ACT_TANH
from jpdse_hip.ops import Act
from jpdse_hip.layers import HipConv2d, InstNormAct, HipResnetBlock
from jpdse_hip.optim import FusedAdam
from oracle.ctu_cpu import nets as onets, model as omodel
from hip_util import DEV, DTYPES, RTOL, to_act, to_nchw, assert_close, bf16_round, quantize_like


def G(seed):
  return torch.Generator().manual_seed(seed)


@pytest.fixture(scope='module', autouse=True)
def _gpu():
  jpdse_hip.require_gpu(0)


# ---- layout ----------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('C', [1, 3, 39, 64])
def test_layout_roundtrip(dtype, C):
  x = torch.randn(2, C, 5, 7, generator=G(C))
  a = to_act(x, dtype)
  assert a.t.shape == (2, 5, 7, (C + 7) // 8 * 8)
  nhwc = a.t.float().cpu()
  assert torch.equal(nhwc[..., :C], quantize_like(x, dtype).permute(0, 2, 3, 1))
  assert (nhwc[..., C:] == 0).all()
  assert torch.equal(to_nchw(a), quantize_like(x, dtype))


# ---- convolution -----------------------------------------------------------------------------
CONV_CASES = [
    # name,           N, H,  W,  C,    K,   k, st, pad, mode,        act
    ('g_first7x7',    2, 12, 20, 39,   64,  7, 1,  3,  PAD_REFLECT, ACT_NONE),
    ('g_down3x3s2',   2, 12, 20, 64,   128, 3, 2,  1,  PAD_ZERO,    ACT_NONE),
    ('g_down_odd',    1, 11, 13, 16,   24,  3, 2,  1,  PAD_ZERO,    ACT_NONE),
    ('resblock1024',  1, 4,  8,  1024, 1024, 3, 1, 1,  PAD_REFLECT, ACT_NONE),
    ('resblock_tail', 3, 5,  9,  72,   136, 3, 1,  1,  PAD_REFLECT, ACT_NONE),
    # 32-pixel-wide images with wide layers: the row-pair form of the nine-tap weight gradient (wgrad_nine.h W32; round 4) --
    # split pixel ranges (16 / 20 tiles), ranges that start inside an image, the first / last row pair of an image
    ('trunk_w32',     2, 6,  32, 256,  256, 3, 1,  1,  PAD_REFLECT, ACT_NONE),
    ('trunk_w32_b',   3, 4,  32, 320,  256, 3, 1,  1,  PAD_REFLECT, ACT_NONE),
    ('trunk_w32_c',   1, 16, 32, 256,  320, 3, 1,  1,  PAD_REFLECT, ACT_NONE),
    ('g_last7x7tanh', 1, 10, 18, 64,   3,   7, 1,  3,  PAD_REFLECT, ACT_TANH),
    ('head64_rows',   1, 16, 128, 64,  3,   7, 1,  3,  PAD_REFLECT, ACT_TANH),    # head_rows_kernel<7, 64>
    ('head32_rows',   2, 24, 256, 32,  3,   7, 1,  3,  PAD_REFLECT, ACT_TANH),    # head_rows_kernel<7, 32> (round 4: LocalEnhancer head)
    ('d_layer0',      2, 16, 24, 39,   64,  4, 2,  2,  PAD_ZERO,    ACT_LRELU),
    ('d_layer2',      2, 9,  13, 128,  256, 4, 2,  2,  PAD_ZERO,    ACT_NONE),
    ('d_layer3_s1',   1, 5,  9,  256,  512, 4, 1,  2,  PAD_ZERO,    ACT_NONE),
    ('d_layer4_1ch',  2, 6,  10, 512,  1,   4, 1,  2,  PAD_ZERO,    ACT_NONE),
    # one-output-channel backward kernels (thin_out1.h): two / three row segments, several items per strip, 128..1024 channels
    ('d_layer4_wide', 2, 9,  130, 512, 1,   4, 1,  2,  PAD_ZERO,    ACT_NONE),
    ('d_layer4_3seg', 3, 5,  300, 128, 1,   4, 1,  2,  PAD_ZERO,    ACT_NONE),
    ('d_layer4_1k',   1, 40, 33, 1024, 1,   4, 1,  2,  PAD_ZERO,    ACT_NONE),
    ('vgg_conv1_1',   1, 16, 24, 3,    64,  3, 1,  1,  PAD_ZERO,    ACT_RELU),
    ('vgg_conv3',     1, 8,  12, 128,  256, 3, 1,  1,  PAD_ZERO,    ACT_RELU),
    ('local_32ch',    1, 14, 22, 39,   32,  7, 1,  3,  PAD_REFLECT, ACT_NONE),
    # M = 33024: 129 tiles of 256 rows x 2 -> the dispatcher picks the 320-row / 2-stage fast tile (208 tiles)
    ('tile320_path',  1, 128, 258, 64, 256, 3, 1,  1,  PAD_ZERO,    ACT_NONE),
    # stream-K wgrad with tiles straddling blocks, 256-row fast fwd tiles, K-tile count 36
    ('fast_256_tiles', 2, 64, 128, 256, 128, 3, 1, 1,  PAD_REFLECT, ACT_NONE),
    # halo kernel: 64-wide N tile, zero pad (fwd and data gradient), 2 slabs; and 128-wide with ReLU epilogue
    ('halo_vgg_64',   2, 8,  64,  128, 64,  3, 1, 1,  PAD_ZERO,    ACT_RELU),
    ('halo_vgg_256',  1, 12, 128, 64,  256, 3, 1, 1,  PAD_ZERO,    ACT_RELU),
    # split-K fast path (few tiles, long reduction): M = 480 -> 2x2 tiles, 36 K-tiles in 2 splits starting
    # inside a tap; reflect (fwd + data gradient on the padded domain) and zero pad with bias + LeakyReLU
    ('splitk_reflect', 1, 15, 32, 256, 256, 3, 1, 1,  PAD_REFLECT, ACT_NONE),
    ('splitk_lrelu',   2, 9,  20, 192, 200, 4, 1, 2,  PAD_ZERO,    ACT_LRELU),
    # thin-input weight gradient (wgrad_thin.h): several 64-pixel strips per row with a ragged last one
    ('thin_ragged',    2, 10, 150, 39, 64,  7, 1, 3,  PAD_REFLECT, ACT_NONE),
    ('thin_s2_k32',    1, 8,  260, 39, 32,  4, 2, 2,  PAD_ZERO,    ACT_LRELU),
    # thin-input forward (thin_fwd.h): stride 2 with 4 rows x 64 px blocks; 48-channel storage -> 4 x 32 px blocks
    ('thinf_s2_40',    2, 20, 150, 39, 64,  4, 2, 2,  PAD_ZERO,    ACT_LRELU),
    ('thinf_s2_48',    1, 18, 100, 42, 64,  4, 2, 2,  PAD_ZERO,    ACT_LRELU),
    ('thinf_s1_48',    1, 13, 100, 42, 64,  7, 1, 3,  PAD_REFLECT, ACT_NONE),
    # head weight gradient (wgrad_thin.h, transposed roles): 32-channel input (LocalEnhancer), ragged strips
    ('head_local32',   2, 9,  70,  32, 3,   7, 1, 3,  PAD_REFLECT, ACT_TANH),
    # round 4: the epilogue staging of the tiled kernels is specialised per activation at compile time -- every copy gets a case
    # on every kernel family (the model itself only asks the wide kernels for none / ReLU)
    ('halo_tanh',      1, 8,  64,  128, 128, 3, 1, 1,  PAD_ZERO,    ACT_TANH),     # gemm_halo_kernel
    ('halo_lrelu',     1, 8,  64,  128, 128, 3, 1, 1,  PAD_REFLECT, ACT_LRELU),
    ('fast_tanh_s2',   2, 16, 48,  128, 128, 3, 2, 1,  PAD_ZERO,    ACT_TANH),     # gemm_fast_kernel (forward), tap program (data gradient)
    ('fast_relu_4x4',  2, 17, 33,  64,  128, 4, 2, 2,  PAD_ZERO,    ACT_RELU),
    ('taps4_tanh',     1, 16, 32,  128, 128, 4, 1, 2,  PAD_ZERO,    ACT_TANH),     # 4x4 stride 1: gemm_taps_kernel core + fringe
    ('thinf_tanh',     1, 16, 64,  39,  64,  7, 1, 3,  PAD_REFLECT, ACT_TANH),     # thin_fwd_kernel
    ('head_3x3_zero',  1, 6,  130, 64, 5,   3, 1, 1,  PAD_ZERO,    ACT_NONE),
    # head forward as a Toeplitz GEMM (4 output pixels x 8 channels per MFMA row): needs OW % 4 == 0
    ('head_toeplitz',  2, 9,  72,  64, 3,   7, 1, 3,  PAD_REFLECT, ACT_TANH),
    ('head_toep_3x3',  1, 5,  36,  24, 7,   3, 1, 1,  PAD_ZERO,    ACT_NONE),
    # per-filter-row weight gradient (wgrad_row.h): 3 taps share the staged dy / input row; stream-K segments
    ('wgrad_row_refl', 1, 6,  64,  128, 256, 3, 1, 1,  PAD_REFLECT, ACT_NONE),
    ('wgrad_row_zero', 2, 5,  128, 256, 256, 3, 1, 1,  PAD_ZERO,    ACT_RELU),
    # all-nine-taps weight gradient (wgrad_nine.h): rolling input-row slots, pixel halves summed through LDS; two column
    # strips + split pixel ranges (slabs), zero padding through the slot of zeros, ragged K / C inside the last tiles,
    # batch boundaries inside a split, a single-row-pair image
    ('nine_strips',    2, 7,  128, 64,  128, 3, 1, 1,  PAD_REFLECT, ACT_NONE),
    ('nine_zero',      3, 5,  64,  128, 64,  3, 1, 1,  PAD_ZERO,    ACT_RELU),
    ('nine_ragged',    1, 9,  64,  120, 200, 3, 1, 1,  PAD_REFLECT, ACT_NONE),
    ('nine_h2',        4, 2,  64,  64,  192, 3, 1, 1,  PAD_REFLECT, ACT_NONE),
    ('nine_wide_nosplit', 1, 4, 64, 512, 512, 3, 1, 1, PAD_ZERO,    ACT_NONE),
    # reflect data gradient = halo kernel on the interior + ring strips (split-K) folded back
    ('ring_dgrad_64',  2, 8,  64,  64,  128, 3, 1, 1,  PAD_REFLECT, ACT_NONE),
    ('ring_dgrad_192', 1, 12, 128, 192, 64,  3, 1, 1,  PAD_REFLECT, ACT_NONE),
    # one-launch form (gemm_halo.h VIRT: the ring rides in the folded frame of dy; >= 128 output channels): two column tiles
    # (left / right edge in different blocks) + interior row tiles + ragged channel tile; a tall single-column image; minimal height
    ('ring_frame_w128', 2, 16, 128, 192, 128, 3, 1, 1, PAD_REFLECT, ACT_NONE),
    ('ring_frame_tall', 1, 24, 64,  128, 256, 3, 1, 1, PAD_REFLECT, ACT_NONE),
    ('ring_frame_h8',   3, 8,  192, 64,  192, 3, 1, 1, PAD_REFLECT, ACT_NONE),
    # all-taps weight gradient (wgrad_taps.h), one case per configuration; ragged 64-pixel chunks, several blocks per tile
    ('taps_3x3s2',     2, 40, 150, 64,  128, 3, 2, 1,  PAD_ZERO,    ACT_NONE),
    ('taps_3x3s1_refl', 2, 36, 100, 64, 64,  3, 1, 1,  PAD_REFLECT, ACT_NONE),
    ('taps_4x4s2',     2, 38, 140, 64,  128, 4, 2, 2,  PAD_ZERO,    ACT_LRELU),
    ('taps_3x3s2_256', 1, 80, 136, 128, 256, 3, 2, 1,  PAD_ZERO,    ACT_NONE),
    # tap-program halo kernel (gemm_taps.h): data gradient of 3x3 stride-2 convs, all four sub-pixel phases from one dy
    # patch; 4 / 8 channel slabs (the ring stage pattern repeats every 3 slabs), one / two 128-wide N tiles, 64-wide tile,
    # several patches per image incl. the zero border right and below, batch 2
    ('tapsprog_128_256', 2, 16, 256, 128, 256, 3, 2, 1, PAD_ZERO,    ACT_NONE),
    ('tapsprog_256_512', 1, 8,  128, 256, 512, 3, 2, 1, PAD_ZERO,    ACT_RELU),
    ('tapsprog_64_256',  1, 8,  128, 64,  256, 3, 2, 1, PAD_ZERO,    ACT_NONE),
    ('tapsprog_128_320', 1, 24, 128, 128, 320, 3, 2, 1, PAD_ZERO,    ACT_NONE),
    # 4x4 stride-1 layers on the tap-program kernel (8 x 32 tiles over the core, 16 taps per slab) + split-K fringe: odd grids
    # with both fringe rectangles (PatchGAN layer 3: 17 x 33 -> 18 x 34 out, core 16 x 32), a core-only grid (forward 16 x 64,
    # its data gradient 15 x 63 has both fringes), 64-wide output tile with bias + LeakyReLU, batch 2, 128 / 256-channel inputs
    ('taps4_l3_small',  2, 17, 33,  256, 512, 4, 1, 2, PAD_ZERO,     ACT_NONE),
    ('taps4_core_only', 1, 15, 63,  128, 128, 4, 1, 2, PAD_ZERO,     ACT_NONE),
    ('taps4_n64_lrelu', 2, 9,  40,  128, 64,  4, 1, 2, PAD_ZERO,     ACT_LRELU),
    # 3x3 stride-1 layers whose grid tiles into 8 x 32 patches but not 4 x 64 (W = 32: the LocalEnhancer trunk): nine-tap
    # program, mirrored / zero padding in the patch loader, split-K over the channel slabs (first case: 2 splits; its data
    # gradient = interior on the same kernel + ring strips + fold), unsplit, 64-wide output tile on a 96-pixel-wide grid
    ('taps9_reflect',   2, 16, 32,  256, 256, 3, 1, 1, PAD_REFLECT,  ACT_NONE),
    ('taps9_zero',      1, 8,  32,  128, 128, 3, 1, 1, PAD_ZERO,     ACT_RELU),
    # reflect data gradient over the folded frame on the 8 x 32 tiles (gemm_taps.h VIRT): three tile rows (an interior one),
    # three column tiles, ragged channel tile
    ('taps9_frame_tall', 1, 24, 32, 128, 192, 3, 1, 1, PAD_REFLECT,  ACT_NONE),
    ('taps9_frame_w96',  2, 16, 96, 128, 128, 3, 1, 1, PAD_REFLECT,  ACT_NONE),
    ('taps9_n64_w96',   1, 8,  96,  192, 64,  3, 1, 1, PAD_ZERO,     ACT_NONE),
    # filter-in-registers row-streaming kernel (conv_rows.h: 64-channel inputs, 3x3, zero pad): stride 2 with 128 / 64
    # outputs (4 x 1 / 2 x 2 waves), stride 1 likewise; several strips, several bands, bands of 16 rows (steady-state
    # look-ahead), borders on every side; the data gradient of the 64-output cases runs on it too
    ('rows_s2_128',    2, 16, 128, 64,  128, 3, 2, 1,  PAD_ZERO,    ACT_NONE),
    ('rows_s2_64',     1, 24, 256, 64,  64,  3, 2, 1,  PAD_ZERO,    ACT_LRELU),
    ('rows_s2_tall',   1, 64, 128, 64,  128, 3, 2, 1,  PAD_ZERO,    ACT_RELU),
    ('rows_s1_64',     2, 12, 128, 64,  64,  3, 1, 1,  PAD_ZERO,    ACT_RELU),
    ('rows_s1_128',    1, 32, 192, 64,  128, 3, 1, 1,  PAD_ZERO,    ACT_NONE),
    # heads as a row-streaming pass (head_rows.h): 7x7 reflect + Tanh forward (bands of 16 rows, 2 strips of 128 pixels,
    # reflected borders on all sides) and the 3x3 zero-pad data gradient of a 3-channel-input conv (VGG conv1_1)
    ('head_rows_7',    2, 48, 256, 64,  3,   7, 1, 3,  PAD_REFLECT, ACT_TANH),
    ('head_rows_7b',   1, 16, 128, 64,  2,   7, 1, 3,  PAD_REFLECT, ACT_NONE),
    ('vgg11_rows',     1, 24, 128, 3,   64,  3, 1, 1,  PAD_ZERO,    ACT_RELU),
    # head data gradient on the padded domain (thin_in_rows.h): interior written in place, ring folded back; several bands
    # and strips (padded width 206 = 3 strips + 14 pixels), one image smaller than a strip
    ('head_dgrad_rows', 2, 40, 200, 64, 3,   7, 1, 3,  PAD_REFLECT, ACT_NONE),
    ('head_dgrad_small', 1, 9,  12,  64, 2,   7, 1, 3,  PAD_REFLECT, ACT_TANH),
]


def _torch_conv(x, w, b, st, pad, mode, act):
  xp = F.pad(x, (pad,) * 4, mode='reflect') if mode == PAD_REFLECT else F.pad(x, (pad,) * 4)
  y = F.conv2d(xp, w, b, stride=st)
  if act == ACT_RELU:
    y = F.relu(y)
  elif act == ACT_LRELU:
    y = F.leaky_relu(y, 0.2)
  elif act == ACT_TANH:
    y = torch.tanh(y)
  return y


def _act_grad_from_output(y, act):
  """d act / d pre-activation as the layer computes it: from the activation OUTPUT (y > 0 <=> pre-activation > 0)."""
  if act == ACT_RELU:
    return (y > 0).to(y.dtype)
  if act == ACT_LRELU:
    return torch.where(y > 0, torch.ones_like(y), torch.full_like(y, 0.2))
  if act == ACT_TANH:
    return 1.0 - y * y
  return torch.ones_like(y)


@pytest.mark.parametrize('seed', [0, 1, 2])
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(case, dtype, seed):
  """Forward, data gradient, weight gradient and bias gradient of one layer against torch-CPU.

  Robust to ReLU / LeakyReLU mask flips instead of seed-tuned (round 1 fixed one seed per case because ~25 % of random
  draws failed): where a pre-activation lies within rounding distance of 0, two correct implementations may take
  different sides, which changes the gradient of that element by O(1).  So the reference backward is evaluated with the
  mask the DEVICE produced (the linear part -- conv data / weight / bias gradient -- is what the kernels compute), and
  the mask itself is checked separately: it may disagree with the reference's only where |pre-activation| is below the
  forward tolerance."""
  name, N, H, W, C, K, k, st, pad, mode, act = case
  g = G(zlib.crc32(name.encode()) % 1000 + 7919 * seed)
  x = quantize_like(torch.randn(N, C, H, W, generator=g), dtype)
  w = torch.randn(K, C, k, k, generator=g) * (1.0 / (C * k * k) ** 0.5)
  b = torch.randn(K, generator=g) * 0.1
  layer = HipConv2d(C, K, k, st, pad, mode, act=act, apply_bias=True, dtype=dtype, device=DEV)
  with torch.no_grad():
    layer.weight.copy_(w)
    layer.bias.copy_(b)
  wq = quantize_like(w, dtype)            # the packed panel is in the compute dtype
  xr = x.clone().requires_grad_(True)
  wr = wq.clone().requires_grad_(True)
  br = b.clone().requires_grad_(True)
  z_ref = _torch_conv(xr, wr, br, st, pad, mode, ACT_NONE)          # pre-activation
  y_ref = _torch_conv(xr, wr, br, st, pad, mode, act).detach()
  gy = quantize_like(torch.randn(y_ref.shape, generator=g), dtype)

  y, ctx = layer.fwd(to_act(x, dtype))
  tol = RTOL[dtype]
  y_dev = to_nchw(y)
  assert_close(y_dev, y_ref, tol, name + ' fwd')
  assert (y.t[..., K:] == 0).all(), 'padding lanes of the output must stay zero'
  # the activation mask of the device vs the reference's: only elements inside the forward tolerance band may differ
  if act in (ACT_RELU, ACT_LRELU):
    flipped = (y_dev > 0) != (z_ref.detach() > 0)
    band = tol * float(y_ref.abs().max())
    assert (z_ref.detach().abs()[flipped] <= band).all(), '%s: activation mask differs outside the rounding band' % name
  z_ref.backward(gy * _act_grad_from_output(y_dev, act))            # reference backward through the device's mask
  dx = layer.bwd(ctx, to_act(gy, dtype), need_dx=True, need_dw=True)
  torch.cuda.synchronize()
  assert_close(to_nchw(dx), xr.grad, tol, name + ' dgrad')
  assert (dx.t[..., C:] == 0).all()
  assert_close(layer.weight.grad.cpu(), wr.grad, tol, name + ' wgrad')
  assert_close(layer.bias.grad.cpu(), br.grad, tol, name + ' bias grad')


@pytest.mark.parametrize('name', ['halo_vgg_64', 'halo_vgg_256', 'ring_dgrad_192'])
def test_halo_kernel_mfma_16x16x32_variant(name):
  # developer mode 19: the halo kernel built on v_mfma_f32_16x16x32_bf16 (other swizzle, fragment and epilogue layout)
  import jpdse_hip
  case = [c for c in CONV_CASES if c[0] == name][0]
  with jpdse_hip.dev_mode(19):          # developer build of the library (include/jpdse_dev.h)
    test_conv_fwd_dgrad_wgrad(case, BF16, 0)


@pytest.mark.parametrize('name', ['taps_4x4s2', 'd_layer2'])
def test_taps_dgrad4_small_shapes(name):
  # developer mode 43: the 4x4 stride-2 data gradient on the tap program (core) + fast kernel (fringe: all eight rectangles
  # for taps_4x4s2) at sizes below the shipped tile-count threshold
  import jpdse_hip
  case = [c for c in CONV_CASES if c[0] == name][0]
  with jpdse_hip.dev_mode(43):
    test_conv_fwd_dgrad_wgrad(case, BF16, 0)


@pytest.mark.parametrize('name', ['d1_like_taps', 'd2_like_taps'])
def test_taps_dgrad4_fused_epilogues(name):
  # the same path with the LeakyReLU mask and the fan-in addend in the core's epilogue and in the fringe's
  import jpdse_hip
  case = [c for c in LRELU_CASES if c[0] == name][0]
  with jpdse_hip.dev_mode(43):
    test_conv_dgrad_fused_lrelu(case, BF16)
  case = [c for c in FUSED_RELU_CASES if c[0] == 'taps_dgrad4_fused'][0]
  with jpdse_hip.dev_mode(43):
    test_conv_dgrad_fused_relu(case, BF16)


# conv -> ReLU(inplace) -> conv chains (VGG19): the second conv's data gradient with the ReLU backward fused
# (jpdse_conv_dgrad_relu), on the halo, fast (merged stride phases), split-K and generic paths
FUSED_RELU_CASES = [
    ('halo_64to128',  1, 8,  64, 64,  128, 3, 1, 1, PAD_ZERO),
    ('halo_128to64',  2, 4,  64, 128, 64,  3, 1, 1, PAD_ZERO),
    ('rows_64to64',   2, 16, 128, 64, 64,  3, 1, 1, PAD_ZERO),
    ('fast_s2',       1, 24, 40, 64,  128, 3, 2, 1, PAD_ZERO),
    ('splitk',        1, 10, 24, 256, 256, 3, 1, 1, PAD_ZERO),
    ('generic_small', 2, 9,  11, 16,  24,  3, 1, 1, PAD_ZERO),
    ('reflect',       1, 12, 20, 64,  64,  3, 1, 1, PAD_REFLECT),
    ('thin1_fused',   2, 7,  70,  256, 1,   4, 1, 2, PAD_ZERO),      # one output channel: the addend rides in thin1_dgrad_kernel
    ('ring_frame',    2, 12, 128, 128, 128, 3, 1, 1, PAD_REFLECT),
    ('taps9_frame_fused', 1, 16, 32, 256, 256, 3, 1, 1, PAD_REFLECT),   # nine-tap program over the frame + split-K finish with addend / mask  # folded-frame halo kernel: addend / mask in its epilogue
    # tap-program kernel epilogues (gemm_taps.h): 4x4 stride 1 (core + split-K fringe: the finish kernel applies the operands
    # there) and the two-set stride-2 data gradient
    ('taps4_fused',   2, 17, 33, 128, 256, 4, 1, 2, PAD_ZERO),
    ('tapsprog_fused', 1, 16, 128, 128, 256, 3, 2, 1, PAD_ZERO),
    ('taps_dgrad4_fused', 2, 17, 131, 64, 128, 4, 2, 2, PAD_ZERO),   # 4x4 stride 2, odd grid: core on the tap program + fringe on the fast kernel
    ('taps9_fused',   1, 16, 32, 256, 256, 3, 1, 1, PAD_ZERO),       # split-K: the finish kernel applies addend / mask
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', FUSED_RELU_CASES, ids=[c[0] for c in FUSED_RELU_CASES])
def test_conv_dgrad_fused_relu(case, dtype):
  name, N, H, W, C, K, k, st, pad, mode = case
  g = G(zlib.crc32(name.encode()) % 1000 + 7)
  z = quantize_like(torch.randn(N, C, H, W, generator=g), dtype)      # pre-activation of the previous layer
  w = torch.randn(K, C, k, k, generator=g) * (1.0 / (C * k * k) ** 0.5)
  layer = HipConv2d(C, K, k, st, pad, mode, act=ACT_NONE, apply_bias=False, dtype=dtype, device=DEV)
  with torch.no_grad():
    layer.weight.copy_(w)
  zr = z.clone().requires_grad_(True)
  wr = quantize_like(w, dtype)
  y_ref = _torch_conv(F.relu(zr), wr, None, st, pad, mode, ACT_NONE)
  gy = quantize_like(torch.randn(y_ref.shape, generator=g), dtype)
  y_ref.backward(gy)
  x = to_act(F.relu(z), dtype)
  y, ctx = layer.fwd(x)
  dz = layer.bwd(ctx, to_act(gy, dtype), need_dx=True, need_dw=False, relu_input=True)
  torch.cuda.synchronize()
  assert_close(to_nchw(dz), zr.grad, RTOL[dtype], name + ' dgrad with fused ReLU mask')
  assert (to_nchw(dz)[z <= 0] == 0).all()
  # gradient fan-in summed in the same epilogue: dz = (dgrad + other) * mask, and without the mask
  other = quantize_like(torch.randn(z.shape, generator=g), dtype)
  dz2 = layer.bwd(ctx, to_act(gy, dtype), need_dx=True, need_dw=False, relu_input=True, addend=to_act(other, dtype))
  assert_close(to_nchw(dz2), zr.grad + other * (z > 0), RTOL[dtype], name + ' dgrad + addend, masked')
  dx3 = layer.bwd(ctx, to_act(gy, dtype), need_dx=True, need_dw=False, addend=to_act(other, dtype))
  y2 = _torch_conv(x_leaf := F.relu(z).clone().requires_grad_(True), wr, None, st, pad, mode, ACT_NONE)
  y2.backward(gy)
  assert_close(to_nchw(dx3), x_leaf.grad + other, RTOL[dtype], name + ' dgrad + addend')


LRELU_CASES = [
    # name, N, H, W, C, K, k, stride, pad, mode -- the discriminator's layer-1 shape family (4x4 stride 2, pad 2, odd
    # extents: merged-phase tile kernel), a split-K small one, and a generic-path one
    ('d1_like',      2, 33, 65, 64, 128, 4, 2, 2, PAD_ZERO),
    ('d1_like_even', 1, 32, 64, 64, 128, 4, 2, 2, PAD_ZERO),
    ('d1_like_taps', 1, 17, 257, 64, 128, 4, 2, 2, PAD_ZERO),      # wide enough for the tap-program core: LeakyReLU mask in its epilogue and in the fringe's
    ('d2_like_taps', 2, 9, 129, 128, 256, 4, 2, 2, PAD_ZERO),
    ('small_splitk', 1, 9, 9, 64, 128, 4, 2, 2, PAD_ZERO),
    ('generic',      1, 11, 13, 24, 40, 3, 1, 1, PAD_ZERO),
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('case', LRELU_CASES, ids=[c[0] for c in LRELU_CASES])
def test_conv_dgrad_fused_lrelu(case, dtype):
  """jpdse_conv_dgrad_fused_lrelu == data gradient (+ addend) followed by the LeakyReLU backward pass, bit for bit,
  and close to autograd through conv(leaky_relu(z))."""
  name, N, H, W, C, K, k, st, pad, mode = case
  g = G(zlib.crc32(name.encode()) % 1000 + 11)
  z = quantize_like(torch.randn(N, C, H, W, generator=g), dtype)
  w = torch.randn(K, C, k, k, generator=g) * (1.0 / (C * k * k) ** 0.5)
  layer = HipConv2d(C, K, k, st, pad, mode, act=ACT_NONE, apply_bias=False, dtype=dtype, device=DEV)
  with torch.no_grad():
    layer.weight.copy_(w)
  zr = z.clone().requires_grad_(True)
  wr = quantize_like(w, dtype)
  y_ref = _torch_conv(F.leaky_relu(zr, 0.2), wr, None, st, pad, mode, ACT_NONE)
  gy = quantize_like(torch.randn(y_ref.shape, generator=g), dtype)
  y_ref.backward(gy)
  xa = to_act(quantize_like(F.leaky_relu(z, 0.2), dtype), dtype)
  y, ctx = layer.fwd(xa)
  dz = layer.bwd(ctx, to_act(gy, dtype), need_dx=True, need_dw=False, relu_input=True, input_slope=0.2)
  dx = layer.bwd(ctx, to_act(gy, dtype), need_dx=True, need_dw=False)
  two_pass = ops.act_bwd(xa, dx, ACT_LRELU, 0.2)
  torch.cuda.synchronize()
  assert torch.equal(dz.t, two_pass.t), name + ': fused LeakyReLU backward differs from the two-pass form'
  assert_close(to_nchw(dz), zr.grad, RTOL[dtype], name + ' dgrad with fused LeakyReLU backward')
  other = quantize_like(torch.randn(z.shape, generator=g), dtype)
  oa = to_act(other, dtype)
  dz2 = layer.bwd(ctx, to_act(gy, dtype), need_dx=True, need_dw=False, relu_input=True, input_slope=0.2, addend=oa)
  dx2 = layer.bwd(ctx, to_act(gy, dtype), need_dx=True, need_dw=False, addend=oa)
  two_pass2 = ops.act_bwd(xa, dx2, ACT_LRELU, 0.2)
  torch.cuda.synchronize()
  assert torch.equal(dz2.t, two_pass2.t), name + ': fused addend + LeakyReLU backward differs from the two-pass form'
  slope = torch.where(z > 0, torch.ones_like(z), torch.full_like(z, 0.2))
  assert_close(to_nchw(dz2), zr.grad + other * slope, RTOL[dtype], name + ' dgrad + addend with fused LeakyReLU backward')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 8, 64, 128, 128), (1, 12, 128, 64, 192), (1, 16, 64, 256, 64), (2, 6, 10, 16, 24), (1, 7, 9, 8, 8)],
                         ids=['halo_128', 'halo_64in_ragged', 'halo_n64', 'generic', 'odd'])
def test_conv_fwd_pool(shape, dtype):
  """jpdse_conv_fwd_pool == jpdse_conv_fwd followed by jpdse_maxpool2_fwd, bit for bit (halo epilogue and the fallback)."""
  N, H, W, C, K = shape
  g = G(N * 1000 + H * 10 + C)
  layer = HipConv2d(C, K, 3, 1, 1, PAD_ZERO, act=ACT_RELU, dtype=dtype, device=DEV)
  with torch.no_grad():
    layer.weight.copy_(torch.randn(K, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5)
    layer.bias.copy_(torch.randn(K, generator=g) * 0.1)
  x = to_act(quantize_like(torch.randn(N, C, H, W, generator=g), dtype), dtype)
  y, ctx = layer.fwd(x)
  ref = ops.maxpool2_fwd(y)
  y2, yp, ctx2 = layer.fwd_pool(x)
  torch.cuda.synchronize()
  assert torch.equal(y.t, y2.t), 'conv output changed by the pooled epilogue'
  assert yp.t.shape == ref.t.shape and torch.equal(yp.t, ref.t), 'pooled output differs from conv + maxpool'


def test_deferred_loss_finals_match_immediate():
  """jpdse_loss_finalize (one launch for all terms) == the per-term second stage, bit for bit; the gradients do not change."""
  g = G(77)
  a = quantize_like(torch.randn(2, 24, 33, 17, generator=g), BF16)
  b = quantize_like(torch.randn(2, 24, 33, 17, generator=g), BF16)
  p = quantize_like(torch.randn(2, 1, 35, 67, generator=g), BF16)
  aa, ba, pa = to_act(a, BF16), to_act(b, BF16), to_act(p, BF16)
  ref = torch.zeros(4, dtype=torch.float32, device=DEV)
  da_ref = ops.l1_fwd_bwd(aa, ba, ref[0:1], 0.7, relu_a=True)
  ops.l1_fwd(aa, ba, ref[1:2])
  ops.mse_fwd(aa, ba, ref[2:3])
  ops.mse_const_fwd(pa, 1.0, ref[3:4])
  out = torch.zeros(4, dtype=torch.float32, device=DEV)
  with ops.deferred_loss_finals():
    da = ops.l1_fwd_bwd(aa, ba, out[0:1], 0.7, relu_a=True)
    ops.l1_fwd(aa, ba, out[1:2])
    ops.mse_fwd(aa, ba, out[2:3])
    ops.mse_const_fwd(pa, 1.0, out[3:4])
    torch.cuda.synchronize()
    assert float(out.abs().sum()) == 0.0, 'slots written before the deferred finalize'
  torch.cuda.synchronize()
  assert torch.equal(out, ref) and torch.equal(da.t, da_ref.t)
  assert abs(float(ref[1]) - float((a - b).abs().mean())) < 1e-5


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('relu_a', [False, True])
def test_l1_fwd_bwd_one_pass(dtype, relu_a):
  g = G(93)
  a = quantize_like(F.relu(torch.randn(2, 24, 9, 7, generator=g)), dtype)
  b = quantize_like(F.relu(torch.randn(2, 24, 9, 7, generator=g)), dtype)
  slot = torch.zeros(1, dtype=torch.float32, device=DEV)
  da = ops.l1_fwd_bwd(to_act(a, dtype), to_act(b, dtype), slot, 2.5, relu_a=relu_a)
  ref = 2.5 * torch.sign(a - b) / a.numel() * ((a > 0) if relu_a else 1.0)
  assert_close(to_nchw(da), ref, RTOL[dtype], 'fused l1 gradient')
  assert abs(slot.item() - (a - b).abs().mean().item()) <= 1e-5 * max(1.0, (a - b).abs().mean().item())


@pytest.mark.parametrize('dtype', DTYPES)
def test_concat_channels(dtype):
  """torch.cat((labels, image), 1) into a base whose label channels are in place (36 + 3 -> 39 of 40 lanes)."""
  g = G(17)
  base = quantize_like(torch.randn(2, 39, 5, 7, generator=g), dtype)
  img = quantize_like(torch.randn(2, 3, 5, 7, generator=g), dtype)
  out = ops.concat_channels(to_act(base, dtype), to_act(img, dtype), 36, to_act(torch.zeros(2, 39, 5, 7), dtype))
  ref = torch.cat((base[:, :36], img), dim=1)
  assert torch.equal(to_nchw(out), ref)
  assert (out.t[..., 39:] == 0).all()


@pytest.mark.parametrize('dtype', DTYPES)
def test_l1_bwd_relu_mask(dtype):
  g = G(91)
  a = quantize_like(F.relu(torch.randn(2, 24, 5, 7, generator=g)), dtype)
  b = quantize_like(F.relu(torch.randn(2, 24, 5, 7, generator=g)), dtype)
  one = torch.ones(1, dtype=torch.float32, device=DEV)
  da = ops.l1_bwd(to_act(a, dtype), to_act(b, dtype), one, 3.0, relu_a=True)
  ref = 3.0 * torch.sign(a - b) / a.numel() * (a > 0)
  assert_close(to_nchw(da), ref, RTOL[dtype], 'l1 backward with ReLU mask')


@pytest.mark.parametrize('dtype', DTYPES)
# (2, 128, 64, 12, 64) and (1, 128, 64, 32, 128): the all-phases row-streaming kernel (dgrad2_rows.h), bands of 4 / 16 dy rows, 1 / 2 strips
# (2, 256, 128, 8, 64) and (1, 512, 256, 4, 128): the tap-program halo kernel (gemm_taps.h) as the forward
@pytest.mark.parametrize('shape', [(2, 128, 64, 5, 7), (1, 1024, 512, 2, 4), (1, 24, 12, 3, 5), (2, 128, 64, 12, 64), (1, 128, 64, 32, 128),
                                   (2, 256, 128, 8, 64), (1, 512, 256, 4, 128)])
def test_conv_transpose(shape, dtype):
  N, Cin, Cout, H, W = shape
  g = G(Cin + H)
  x = quantize_like(torch.randn(N, Cin, H, W, generator=g), dtype)
  w = torch.randn(Cin, Cout, 3, 3, generator=g) * (1.0 / (Cin * 9) ** 0.5)
  layer = HipConv2d(Cin, Cout, 3, 2, 1, transposed=True, apply_bias=False, dtype=dtype, device=DEV)
  with torch.no_grad():
    layer.weight.copy_(w)
  xr = x.clone().requires_grad_(True)
  wr = quantize_like(w, dtype).clone().requires_grad_(True)
  y_ref = F.conv_transpose2d(xr, wr, None, stride=2, padding=1, output_padding=1)
  gy = quantize_like(torch.randn(y_ref.shape, generator=g), dtype)
  y_ref.backward(gy)
  y, ctx = layer.fwd(to_act(x, dtype))
  assert y.t.shape[1:3] == (2 * H, 2 * W)
  tol = RTOL[dtype]
  assert_close(to_nchw(y), y_ref.detach(), tol, 'convT fwd')
  dx = layer.bwd(ctx, to_act(gy, dtype), True, True)
  assert_close(to_nchw(dx), xr.grad, tol, 'convT dgrad')
  assert_close(layer.weight.grad.cpu(), wr.grad, tol, 'convT wgrad')


# PatchGAN layer 0, data gradient with respect to the 3 image channels only (HipConv2d.bwd_input_slice): the row-streaming
# kernel of thin_dgrad2_rows.h (dx width a multiple of 256) and the stride-phase path (any other width)
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('shape', [(2, 16, 256), (1, 24, 512), (1, 10, 36)])
def test_conv_dgrad_input_slice(shape, dtype):
  N, H, W = shape
  g = G(N * H + W)
  x = quantize_like(torch.randn(N, 39, H, W, generator=g), dtype)
  w = torch.randn(64, 39, 4, 4, generator=g) * (1.0 / (39 * 16) ** 0.5)
  layer = HipConv2d(39, 64, 4, 2, 2, PAD_ZERO, act=ACT_LRELU, apply_bias=True, dtype=dtype, device=DEV)
  with torch.no_grad():
    layer.weight.copy_(w)
  xr = x.clone().requires_grad_(True)
  wq = quantize_like(w, dtype)
  z_ref = F.conv2d(F.pad(xr, (2,) * 4), wq, layer.bias.detach().cpu(), stride=2)
  y, ctx = layer.fwd(to_act(x, dtype))
  y_dev = to_nchw(y)
  gy = quantize_like(torch.randn(z_ref.shape, generator=g), dtype)
  z_ref.backward(gy * _act_grad_from_output(y_dev, ACT_LRELU))      # through the device's activation mask (see above)
  dx = layer.bwd_input_slice(ctx, to_act(gy, dtype), 36, 39)
  torch.cuda.synchronize()
  assert dx.C == 3 and dx.t.shape[1:3] == (H, W)
  assert_close(to_nchw(dx), xr.grad[:, 36:39], RTOL[dtype], 'layer-0 data gradient, image channels')
  assert (dx.t[..., 3:] == 0).all()


# conv -> InstanceNorm with the norm's moment pass fused into the conv epilogue (jpdse_conv_fwd_moments): the three producers
# that have the epilogue -- thin_fwd (first 7x7 conv), conv_rows (64 -> 128 stride 2), dgrad2_rows (ConvTranspose 128 -> 64)
# -- against the unfused sequence on the same layer (same kernels, moments from a separate pass) and against torch.
from jpdse_hip.layers import ConvNormAct

FUSED_MOMENT_CASES = [
    ('first7x7',  lambda: HipConv2d(39, 64, 7, 1, 3, PAD_REFLECT, apply_bias=False, dtype=BF16, device=DEV), (2, 39, 16, 128)),
    ('down_s2',   lambda: HipConv2d(64, 128, 3, 2, 1, PAD_ZERO, apply_bias=False, dtype=BF16, device=DEV), (2, 64, 32, 128)),
    ('convT_up',  lambda: HipConv2d(128, 64, 3, 2, 1, transposed=True, apply_bias=False, dtype=BF16, device=DEV), (2, 128, 16, 64)),
    # halo kernel epilogue (ResnetBlock convs) + the one-kernel norm that merges its slots (inorm_reg_fwd_kernel PHASE 3)
    ('resblock_halo', lambda: HipConv2d(128, 128, 3, 1, 1, PAD_REFLECT, apply_bias=False, dtype=BF16, device=DEV), (2, 128, 8, 128)),
]


@pytest.mark.parametrize('name,make,shape', FUSED_MOMENT_CASES, ids=[c[0] for c in FUSED_MOMENT_CASES])
def test_conv_norm_fused_moments(name, make, shape):
  g = G(zlib.crc32(name.encode()) & 0xfff)
  conv = make()
  with torch.no_grad():
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (1.0 / (conv.weight[0].numel()) ** 0.5))
  stage = ConvNormAct(conv, InstNormAct(ACT_RELU))
  x = to_act(quantize_like(torch.randn(*shape, generator=g), BF16), BF16)
  assert conv.fwd_moments(x) is not None, 'this layer is expected to take the fused epilogue'
  y, ctx = stage.fwd(x)
  gy = to_act(quantize_like(torch.randn(to_nchw(y).shape, generator=g), BF16), BF16)
  dx = stage.bwd(ctx, gy, True, True)
  dw = conv.weight.grad.clone()
  stats = ctx.items[1].items[1].clone()
  torch.cuda.synchronize()
  # unfused reference on the same layer
  conv.fwd_moments = lambda _x: None
  y2, ctx2 = stage.fwd(x)
  dx2 = stage.bwd(ctx2, gy, True, True)
  torch.cuda.synchronize()
  C = to_nchw(y).shape[1]
  assert torch.equal(ctx.items[1].items[0].t, ctx2.items[1].items[0].t), 'the conv output itself must not change'
  assert_close(stats.cpu()[:, :C], ctx2.items[1].items[1].cpu()[:, :C], 2e-5, name + ' stats (per-block (mean, M2) of the stored values, Chan merge) vs the separate shifted pass')
  assert_close(to_nchw(y), to_nchw(y2), RTOL[BF16], name + ' norm output')
  assert_close(to_nchw(dx), to_nchw(dx2), RTOL[BF16], name + ' dx')
  assert_close(dw.cpu(), conv.weight.grad.cpu(), RTOL[BF16], name + ' dw')
  # and against torch on the fp32 view of the same conv output
  h = to_nchw(ctx.items[1].items[0])
  assert_close(to_nchw(y), F.relu(F.instance_norm(h, eps=1e-5)), RTOL[BF16], name + ' vs torch')


FUSED_MOMENT_FULL = [
    ('first7x7@1024x512', lambda: HipConv2d(39, 64, 7, 1, 3, PAD_REFLECT, apply_bias=False, dtype=BF16, device=DEV), (2, 39, 512, 1024)),
    ('down_s2@1024x512',  lambda: HipConv2d(64, 128, 3, 2, 1, PAD_ZERO, apply_bias=False, dtype=BF16, device=DEV), (2, 64, 512, 1024)),
    ('convT_up@1024x512', lambda: HipConv2d(128, 64, 3, 2, 1, transposed=True, apply_bias=False, dtype=BF16, device=DEV), (2, 128, 256, 512)),
    ('resblock@1024x512', lambda: HipConv2d(1024, 1024, 3, 1, 1, PAD_REFLECT, apply_bias=False, dtype=BF16, device=DEV), (4, 1024, 32, 64)),
]


@pytest.mark.parametrize('name,make,shape', FUSED_MOMENT_FULL, ids=[c[0] for c in FUSED_MOMENT_FULL])
def test_conv_norm_fused_moments_offset_channels_full_size(name, make, shape):
  """channels whose mean is tens of standard deviations away from zero (a one-hot label plane times a filter with
  a non-zero sum does that), at the bench resolution (HW = 524288 per channel): the fused moments -- per-block (mean, M2)
  about a pilot, Chan merge -- must give the statistics of the separate, shifted moment pass.  Inputs 4 + N(0,1), filter
  N(0.01, 0.02): channel means ~ +-20 .. 40 sigma."""
  g = G(zlib.crc32(name.encode()) & 0xfff)
  conv = make()
  with torch.no_grad():
    wmean = torch.full(conv.weight.shape, 0.01)
    if conv.transposed:
      # the sub-pixel phases of a stride-2 ConvTranspose use 1 or 2 filter rows / columns: halve the mean of the taps that
      # come in pairs so that every phase has the same channel mean (otherwise the phases, not the noise, set the variance)
      half = torch.tensor([0.5, 1.0, 0.5])
      wmean = wmean * half.view(1, 1, 3, 1) * half.view(1, 1, 1, 3)
    # (for the ConvTranspose also a smaller random part: the random weights of a phase sum to a per-phase constant, and
    # those constants differing between the four phases is spatial variance of the channel)
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (0.002 if conv.transposed else 0.02) + wmean)
  stage = ConvNormAct(conv, InstNormAct(ACT_RELU))
  N, C, H, W = shape
  xd = torch.randn((N, H, W, C), generator=torch.Generator(device=DEV).manual_seed(7), device=DEV) + 4.0
  x = Act.empty(N, H, W, C, BF16, DEV)
  x.t.zero_()
  x.t[..., :C] = xd.to(torch.bfloat16)
  del xd
  assert conv.fwd_moments(x) is not None, 'this layer is expected to take the fused epilogue'
  y, ctx = stage.fwd(x)
  stats = ctx.items[1].items[1].clone()
  h = ctx.items[1].items[0]
  K = h.C
  hf = h.t[..., :K].float()
  mean_ref = hf.mean(dim=(1, 2)).double()
  var_ref = hf.double().var(dim=(1, 2), unbiased=False)
  ratio = (mean_ref.abs() / var_ref.sqrt()).min().item()
  assert ratio >= 10.0, 'test premise: every channel at least 10 sigma off zero (got %.1f)' % ratio
  conv.fwd_moments = lambda _x: None          # the separate (shifted) moment pass on the same conv output
  y2, ctx2 = stage.fwd(x)
  torch.cuda.synchronize()
  stats2 = ctx2.items[1].items[1]
  assert torch.equal(h.t, ctx2.items[1].items[0].t)
  rstd_ref = (var_ref + 1e-5).rsqrt()
  assert_close(stats[:, :K, 0].cpu(), mean_ref.cpu(), 1e-6, name + ' fused mean vs fp64 of the stored tensor')
  assert_close(stats[:, :K, 1].cpu(), rstd_ref.cpu(), 5e-5, name + ' fused rstd vs fp64 of the stored tensor')
  assert_close(stats2[:, :K, 1].cpu(), rstd_ref.cpu(), 5e-5, name + ' separate-pass rstd vs fp64 of the stored tensor')
  assert_close(y.t.float().cpu(), y2.t.float().cpu(), RTOL[BF16], name + ' norm output, fused vs separate moments')


# Persistent, cross-tile-pipelined fast kernel (gemm_pers.h; round 4): forward and merged-phase data gradient of the short-K
# layers.  It is NOT on the default dispatch any more (once the epilogue staging of gemm_fast_kernel was fixed that kernel became
# the faster one); it stays built and tested: developer mode 52 sends small problems
# through it -- ragged M tails, ragged N tails, tiles that cross image boundaries, blocks with several tiles and blocks
# with one -- against torch-CPU and against the kernels the default dispatch takes for the same layer (mode 50).
PERS_CASES = [
    # name,          N, H,  W,   C,   K,   k, st, pad, mode
    ('d1_like',      3, 33, 65,  64,  128, 4, 2,  2,   PAD_ZERO),
    ('d2_like',      2, 17, 33,  128, 256, 4, 2,  2,   PAD_ZERO),
    ('down_128_256', 2, 24, 40,  128, 256, 3, 2,  1,   PAD_ZERO),
    ('n_tail_192',   1, 21, 37,  64,  192, 3, 2,  1,   PAD_ZERO),       # N tail: 192 = 128 + 64 channels
    ('narrow_64',    2, 19, 30,  128, 64,  3, 1,  1,   PAD_ZERO),       # 128 x 64 tiles
    ('reflect_s1',   1, 18, 26,  64,  128, 3, 1,  1,   PAD_REFLECT),
    ('many_tiles',   4, 130, 254, 64, 128, 4, 2,  2,   PAD_ZERO),       # 264 tiles on 256 CUs: blocks with one tile and blocks with two
]


@pytest.mark.parametrize('case', PERS_CASES, ids=[c[0] for c in PERS_CASES])
def test_persistent_fast_kernel(case):
  name, N, H, W, C, K, k, st, pad, mode = case
  g = G(zlib.crc32(name.encode()) % 1000 + 3)
  x = quantize_like(torch.randn(N, C, H, W, generator=g), BF16)
  w = torch.randn(K, C, k, k, generator=g) * (1.0 / (C * k * k) ** 0.5)
  xr = x.clone().requires_grad_(True)
  y_ref = _torch_conv(xr, quantize_like(w, BF16), None, st, pad, mode, ACT_NONE)
  gy = quantize_like(torch.randn(y_ref.shape, generator=g), BF16)
  y_ref.backward(gy)
  out = {}
  for fm in (52, 50):
    with jpdse_hip.dev_mode(fm):
      layer = HipConv2d(C, K, k, st, pad, mode, act=ACT_NONE, apply_bias=False, dtype=BF16, device=DEV)
      with torch.no_grad():
        layer.weight.copy_(w)
      y, ctx = layer.fwd(to_act(x, BF16))
      dx = layer.bwd(ctx, to_act(gy, BF16), need_dx=True, need_dw=False)
      torch.cuda.synchronize()
      out[fm] = (y.t.clone(), dx.t.clone())
  assert_close(to_nchw(Act(out[52][0], K)), y_ref.detach(), RTOL[BF16], name + ' fwd (persistent kernel)')
  assert_close(to_nchw(Act(out[52][1], C)), xr.grad, RTOL[BF16], name + ' dgrad (persistent kernel)')
  assert (out[52][0][..., K:] == 0).all() and (out[52][1][..., C:] == 0).all(), 'padding lanes must stay zero'
  assert_close(out[52][0].float().cpu(), out[50][0].float().cpu(), RTOL[BF16], name + ' fwd: persistent kernel vs the default dispatch')


# InstanceNorm backward with its two per-channel sums taken from the epilogue of the data-gradient kernel that produced dy
# (jpdse_conv_dgrad_fused_nsums -> jpdse_inorm_bwd_from_sums; round 4): the ResnetBlock chain on the folded-frame halo kernel.
@pytest.mark.parametrize('shape', [(2, 128, 8, 128), (1, 256, 16, 64)], ids=['128ch', '256ch'])
def test_resblock_norm_backward_sums_from_dgrad_epilogue(shape):
  from jpdse_hip.layers import HipResnetBlock, run_chain_fwd, run_chain_bwd
  N, C, H, W = shape
  g = G(C + H)
  blocks = [HipResnetBlock(C, dtype=BF16, device=DEV) for _ in range(3)]
  with torch.no_grad():
    for b in blocks:
      for i in (1, 5):
        b.conv_block[i].weight.copy_(torch.randn(b.conv_block[i].weight.shape, generator=g) * (1.0 / (C * 9)) ** 0.5)
  x = to_act(quantize_like(torch.randn(N, C, H, W, generator=g), BF16), BF16)
  y, ctxs = run_chain_fwd(blocks, x)
  gy = to_act(quantize_like(torch.randn(N, C, H, W, generator=g), BF16), BF16)
  calls = []
  orig = ops.conv_dgrad_nsums
  ops.conv_dgrad_nsums = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
  try:
    dx = run_chain_bwd(blocks, ctxs, gy, True, True)
  finally:
    ops.conv_dgrad_nsums = orig
  torch.cuda.synchronize()
  # block 2: conv2 -> norm1 sums, conv1 -> block 1's norm2 sums; block 1 likewise; block 0: conv2 only (its dx leaves the chain)
  assert len(calls) == 5, 'expected 5 data gradients with the norm-backward-sum epilogue, saw %d' % len(calls)
  dws = [b.conv_block[i].weight.grad.clone() for b in blocks for i in (1, 5)]
  for b in blocks:
    b.fuse_norm_sums = False
  dx2 = run_chain_bwd(blocks, ctxs, gy, True, True)
  torch.cuda.synchronize()
  assert_close(to_nchw(dx), to_nchw(dx2), RTOL[BF16], 'ResnetBlock chain dx: norm sums from the dgrad epilogue vs the separate pass')
  for a, b in zip(dws, [b.conv_block[i].weight.grad for b in blocks for i in (1, 5)]):
    assert_close(a.cpu(), b.cpu(), RTOL[BF16], 'ResnetBlock chain dw: norm sums from the dgrad epilogue vs the separate pass')


def test_dgrad_nsums_slots_vs_fp64():
  """The slots themselves: sum dz and sum dz * yhat per (image, channel) against an fp64 evaluation from the tensors the
  kernel stored / read (dy as written, x, stats), ReLU norm and plain norm."""
  from jpdse_hip.layers import InstNormAct
  N, C, H, W = 2, 128, 12, 128
  g = G(91)
  conv = HipConv2d(C, C, 3, 1, 1, PAD_REFLECT, apply_bias=False, dtype=BF16, device=DEV)
  with torch.no_grad():
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (1.0 / (C * 9)) ** 0.5)
  xin = to_act(quantize_like(torch.randn(N, C, H, W, generator=g), BF16), BF16)
  y, ctx = conv.fwd(xin)
  gy = to_act(quantize_like(torch.randn(N, C, H, W, generator=g), BF16), BF16)
  for act in (ACT_RELU, ACT_NONE):
    norm = InstNormAct(act)
    nx = to_act(quantize_like(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.5, BF16), BF16)      # the norm's input
    _, nctx = norm.fwd(nx)
    other = to_act(quantize_like(torch.randn(N, C, H, W, generator=g), BF16), BF16)
    dx = conv.bwd(ctx, gy, need_dx=True, need_dw=False, addend=other, sink=norm.sink(nctx))
    ref = conv.bwd(ctx, gy, need_dx=True, need_dw=False, addend=other)
    torch.cuda.synchronize()
    assert dx.nsums is not None and torch.equal(dx.t, ref.t), 'the data gradient itself must not change'
    sums, slots, _ = dx.nsums
    assert slots == (H // 4) * (W // 64) and tuple(sums.shape) == (N, C, slots, 2)
    stats = nctx.items[1].double()
    xv, dv = nx.t[..., :C].double(), dx.t[..., :C].double()
    yh = (xv - stats[:, None, None, :C, 0]) * stats[:, None, None, :C, 1]
    dz = dv * (yh > 0).double() if act == ACT_RELU else dv
    want = torch.stack([dz.sum(dim=(1, 2)), (dz * yh).sum(dim=(1, 2))], dim=-1)
    got = sums.double().sum(dim=2)
    scale = (dz.pow(2).sum(dim=(1, 2)).sqrt()[..., None] * torch.stack([torch.ones_like(want[..., 0]).mul(H * W).sqrt(), yh.pow(2).sum(dim=(1, 2)).sqrt()], dim=-1))
    err = ((got - want).abs() / scale).max().item()
    from hip_util import record
    record('norm-backward sums from the dgrad epilogue vs fp64 (of the Cauchy-Schwarz scale), act %d' % act, err, 2e-6)
    assert err <= 2e-6, err
    # and the backward through them equals the separate-pass backward
    a = norm.bwd(nctx, dx)
    dx.nsums = None
    b = norm.bwd(nctx, dx)
    torch.cuda.synchronize()
    assert_close(to_nchw(a), to_nchw(b), RTOL[BF16], 'inorm_bwd_from_sums vs inorm_bwd, act %d' % act)


def test_prof_hbm_reselect_with_a_smaller_maximum():
  """jpdse_prof_hbm_select(1, big) then (1, small): the region count must be bounded by the CURRENT maximum (the
  event vector only grows; the byte / class vectors are re-sized), so no region index runs past them."""
  import ctypes
  L = jpdse_hip.lib()
  x = to_act(torch.randn(1, 16, 8, 8), F32)
  for cap, calls in ((6, 9), (2, 7), (3, 1)):
    jpdse_hip.check(L.jpdse_prof_hbm_select(1, cap), 'prof_hbm_select')
    for _ in range(calls):
      ops.inorm_fwd(x, ACT_NONE)
    torch.cuda.synchronize()
    ms, by, n = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    jpdse_hip.check(L.jpdse_prof_hbm_collect(0, ctypes.byref(ms), ctypes.byref(by), ctypes.byref(n)), 'prof_hbm_collect')
    assert n.value == min(cap, calls) and ms.value > 0.0 and by.value > 0.0, (cap, calls, n.value)
  jpdse_hip.check(L.jpdse_prof_hbm_select(0, 0), 'prof_hbm_select off')


# ---- instance norm -----------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('act', [ACT_NONE, ACT_RELU, ACT_LRELU])
@pytest.mark.parametrize('shape', [(2, 64, 9, 13), (1, 1024, 4, 8), (3, 40, 16, 24), (1, 8, 33, 65)])
def test_instance_norm_act(shape, act, dtype):
  N, C, H, W = shape
  g = G(C * H + act)
  x = quantize_like(torch.randn(N, C, H, W, generator=g) * 2.0 + 3.0, dtype)   # mean >> 0: stresses the variance
  xr = x.clone().requires_grad_(True)
  y_ref = F.instance_norm(xr, eps=1e-5)
  if act == ACT_RELU:
    y_ref = F.relu(y_ref)
  elif act == ACT_LRELU:
    y_ref = F.leaky_relu(y_ref, 0.2)
  gy = quantize_like(torch.randn(y_ref.shape, generator=g), dtype)
  y_ref.backward(gy)
  layer = InstNormAct(act)
  y, ctx = layer.fwd(to_act(x, dtype))
  tol = RTOL[dtype]
  assert_close(to_nchw(y), y_ref.detach(), tol, 'inorm fwd')
  dx = layer.bwd(ctx, to_act(gy, dtype))
  assert_close(to_nchw(dx), xr.grad, tol, 'inorm bwd')



# register-held InstanceNorm forms (norm.hip, inorm_reg_*): shapes with several pixel splits per (image, channel block), a
# ragged last split, a ragged channel block, both register depths (8 / 16 pixels per thread), and one tensor with too many
# splits (three-kernel form).  Checked against torch, against the three-kernel form and the one-kernel form (in-launch
# exchange) of the developer build, and for run-to-run bit equality (partial rows are summed in split order).
FUSED_NORM_SHAPES = [
    ('resblock_real', (4, 1024, 32, 64), True),    # the 36 ResnetBlock norms of the bench step: 512 blocks, 16 splits
    ('ragged_cols',   (2, 72, 40, 50), False),     # 9 channel vectors in a 16-wide block, 16 splits of 128 pixels
    ('d_scale2_p16',  (8, 128, 65, 129), False),   # 8385 pixels: 33 splits of 256 (16 pixels per thread), ragged tail
    ('down3_p16',     (2, 512, 64, 128), False),
    ('too_large',     (1, 64, 512, 512), False),   # 512 splits: stays on moment -> finalize -> apply
]


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('name,shape,with_res', FUSED_NORM_SHAPES, ids=[c[0] for c in FUSED_NORM_SHAPES])
def test_instance_norm_exchange(name, shape, with_res, dtype):
  N, C, H, W = shape
  g = G(zlib.crc32(name.encode()) & 0xffff)
  x = quantize_like(torch.randn(N, C, H, W, generator=g) * 1.5 + 0.7, dtype)
  res = quantize_like(torch.randn(N, C, H, W, generator=g), dtype) if with_res else None
  gy = quantize_like(torch.randn(N, C, H, W, generator=g), dtype)
  act = ACT_NONE if with_res else ACT_RELU
  xr = x.clone().requires_grad_(True)
  y_ref = F.instance_norm(xr, eps=1e-5)
  y_ref = y_ref + res if with_res else F.relu(y_ref)
  y_ref.backward(gy)
  xa, ga = to_act(x, dtype), to_act(gy, dtype)
  ra = to_act(res, dtype) if with_res else None

  def run():
    y, stats = ops.inorm_fwd(xa, act, 0.2, 1e-5, ra)
    dx = ops.inorm_bwd(xa, stats, ga, act, 0.2, 1e-5)
    torch.cuda.synchronize()
    return y, stats, dx

  y, stats, dx = run()
  tol = RTOL[dtype]
  assert_close(to_nchw(y), y_ref.detach(), tol, 'inorm fwd')
  assert_close(to_nchw(dx), xr.grad, tol, 'inorm bwd')
  for rep in range(3):                      # every launch uses a fresh epoch of the exchange flags
    y2, stats2, dx2 = run()
    assert torch.equal(y2.t, y.t) and torch.equal(stats2, stats) and torch.equal(dx2.t, dx.t), 'not reproducible'
  with jpdse_hip.dev_mode(27):              # developer build: always the three-kernel form
    y3, stats3, dx3 = run()
  # same math, different summation order: fp32 sums of <= 8385 terms
  st_tol = 2e-5
  assert_close(stats3.cpu()[:, :C], stats.cpu()[:, :C], st_tol, 'stats vs three-kernel form')
  assert_close(to_nchw(y3), to_nchw(y), tol, 'fwd vs three-kernel form')
  assert_close(to_nchw(dx3), to_nchw(dx), tol, 'bwd vs three-kernel form')
  with jpdse_hip.dev_mode(28):              # one kernel, rows exchanged inside the launch (8 or 16 pixels per thread: its own
    y4, stats4, dx4 = run()                 # split count, so another summation order than the shipped form)
    assert_close(stats4.cpu()[:, :C], stats.cpu()[:, :C], st_tol, 'stats vs one-kernel form')
    assert_close(to_nchw(y4), to_nchw(y), tol, 'fwd vs one-kernel form')
    assert_close(to_nchw(dx4), to_nchw(dx), tol, 'bwd vs one-kernel form')
    for rep in range(3):
      y5, stats5, dx5 = run()
      assert torch.equal(y5.t, y4.t) and torch.equal(stats5, stats4) and torch.equal(dx5.t, dx4.t), 'one-kernel form not reproducible'

@pytest.mark.parametrize('dtype', DTYPES)
def test_resnet_block(dtype):
  dim, H, W = 64, 6, 10
  g = G(77)
  sd = {'b.conv_block.1.weight': torch.randn(dim, dim, 3, 3, generator=g) * 0.05,
        'b.conv_block.1.bias': torch.randn(dim, generator=g),
        'b.conv_block.5.weight': torch.randn(dim, dim, 3, 3, generator=g) * 0.05,
        'b.conv_block.5.bias': torch.randn(dim, generator=g)}
  x = quantize_like(torch.randn(2, dim, H, W, generator=g), dtype)
  blk = HipResnetBlock(dim, dtype=dtype, device=DEV)
  blk.load_state_dict({k[2:]: v for k, v in sd.items()})
  ref_sd = {k: (quantize_like(v, dtype) if k.endswith('weight') else v).clone().requires_grad_(True)
            for k, v in sd.items()}
  xr = x.clone().requires_grad_(True)
  y_ref = onets.resblock(ref_sd, 'b', xr)
  gy = quantize_like(torch.randn(y_ref.shape, generator=g), dtype)
  y_ref.backward(gy)
  y, ctx = blk.fwd(to_act(x, dtype))
  tol = RTOL[dtype]
  assert_close(to_nchw(y), y_ref.detach(), tol, 'resblock fwd')
  dx = blk.bwd(ctx, to_act(gy, dtype), True, True)
  assert_close(to_nchw(dx), xr.grad, tol, 'resblock dx')
  assert_close(blk.conv_block[1].weight.grad.cpu(), ref_sd['b.conv_block.1.weight'].grad, tol, 'resblock dw1')
  assert_close(blk.conv_block[5].weight.grad.cpu(), ref_sd['b.conv_block.5.weight'].grad, tol, 'resblock dw5')


# ---- pooling -----------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('hw', [(8, 12), (9, 13), (1, 5), (2, 2)])
def test_avgpool3s2(hw, dtype):
  H, W = hw
  x = quantize_like(torch.randn(2, 39, H, W, generator=G(H * W)), dtype)
  xr = x.clone().requires_grad_(True)
  y_ref = onets.avgpool3s2(xr)
  gy = quantize_like(torch.randn(y_ref.shape, generator=G(1)), dtype)
  y_ref.backward(gy)
  a = to_act(x, dtype)
  y = ops.avgpool3s2_fwd(a)
  assert_close(to_nchw(y), y_ref.detach(), RTOL[dtype], 'avgpool fwd')
  dx = ops.avgpool3s2_bwd(to_act(gy, dtype), H, W)
  assert_close(to_nchw(dx), xr.grad, RTOL[dtype], 'avgpool bwd')


@pytest.mark.parametrize('dtype', DTYPES)
@pytest.mark.parametrize('hw', [(8, 12), (9, 13)])
def test_maxpool2(hw, dtype):
  H, W = hw
  x = quantize_like(torch.randn(2, 64, H, W, generator=G(H)), dtype)
  xr = x.clone().requires_grad_(True)
  y_ref = F.max_pool2d(xr, 2, 2)
  gy = quantize_like(torch.randn(y_ref.shape, generator=G(2)), dtype)
  y_ref.backward(gy)
  a = to_act(x, dtype)
  assert torch.equal(to_nchw(ops.maxpool2_fwd(a)), y_ref.detach())
  dx = ops.maxpool2_bwd(a, to_act(gy, dtype))
  assert torch.equal(to_nchw(dx), xr.grad)


# ---- elementwise -------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', DTYPES)
def test_act_bwd_add_channel_ops(dtype):
  g = G(9)
  y = quantize_like(torch.randn(2, 24, 5, 7, generator=g), dtype)
  dy = quantize_like(torch.randn(2, 24, 5, 7, generator=g), dtype)
  ya, da = to_act(y, dtype), to_act(dy, dtype)
  tol = RTOL[dtype]
  assert_close(to_nchw(ops.act_bwd(ya, da, ACT_RELU)), dy * (y > 0).float(), tol, 'relu bwd')
  assert_close(to_nchw(ops.act_bwd(ya, da, ACT_LRELU, 0.2)), dy * torch.where(y > 0, 1.0, 0.2), tol, 'lrelu bwd')
  yt = quantize_like(torch.tanh(y), dtype)
  assert_close(to_nchw(ops.act_bwd(to_act(yt, dtype), da, ACT_TANH)), dy * (1 - yt * yt), tol, 'tanh bwd')
  assert_close(to_nchw(ops.add(ya, da)), y + dy, tol, 'add')
  out = torch.zeros(24, device=DEV)
  ops.channel_sum(da, out)
  assert_close(out.cpu(), dy.sum((0, 2, 3)), tol, 'channel_sum')
  dst = Act(torch.zeros(2, 5, 7, 40, dtype=ya.t.dtype, device=DEV), 39)
  ops.channel_copy(ya, 0, dst, 36, 3)
  got = to_nchw(dst)
  assert torch.equal(got[:, 36:39], y[:, 0:3]) and (got[:, :36] == 0).all()


def test_onehot_edge_integer_exact(golden_dir):
  import os
  gd = np.load(os.path.join(golden_dir, 'preprocess_cityscapes_crop.npz'))
  lab = torch.tensor(gd['label'].astype(np.float32))[None, None]
  lab[lab == 255] = 35
  ins = torch.tensor(gd['instance'].astype(np.int64))[None, None]
  for dtype in DTYPES:
    out = ops.onehot_edge(lab.to(DEV).contiguous(), ins.to(DEV).contiguous(), 35, 39, dtype)
    got = to_nchw(out)
    assert np.array_equal(got[:, :36].numpy().astype(np.uint8), gd['input_label'])   # the reference's output
    assert (got[:, 36:] == 0).all()
  # batch > 1, ragged size, ids up to 34, vs the oracle
  xd = omodel.synthetic_batch(3, 20, 44, seed=5)
  ref = omodel.preprocess(xd, omodel.default_opt())
  out = ops.onehot_edge(xd['label'].to(DEV).contiguous(), xd['instance'].to(DEV).contiguous(), 35, 39, F32)
  assert torch.equal(to_nchw(out)[:, :36], ref)


@pytest.mark.parametrize('dtype', DTYPES)
def test_input_builder_one_pass_equals_onehot_edge_plus_concats(dtype, golden_dir):
  """jpdse_input_builder (generator input + both discriminator-input halves from the label / instance maps in one pass) and
  jpdse_insert_channels (the generated image into the fake half) against the separate kernels they replace, bit for bit --
  and through them against the reference's preprocess output on the Cityscapes crop (integer-exact, model.py:375-394)."""
  import os
  xd = omodel.synthetic_batch(3, 20, 44, seed=5)
  lab, ins = xd['label'].to(DEV).contiguous(), xd['instance'].to(DEV).contiguous()
  g = G(3)
  imgs = [to_act(quantize_like(torch.randn(3, 3, 20, 44, generator=g), dtype), dtype) for _ in range(3)]
  base = ops.onehot_edge(lab, ins, 35, 39, dtype)
  want = [ops.concat_channels(base, im, 36, base.empty_like()) for im in imgs]
  dsts = [base.empty_like() for _ in range(3)]
  for d in dsts:
    d.t.fill_(7.0)                                        # every lane must be written
  ops.input_builder(lab, ins, 35, dsts, [imgs[0], imgs[1], None], 36)
  assert torch.equal(dsts[0].t, want[0].t) and torch.equal(dsts[1].t, want[1].t)
  assert torch.equal(dsts[2].t[..., :36], base.t[..., :36]) and (dsts[2].t[..., 36:] == 0).all()
  ops.insert_channels(dsts[2], imgs[2], 36)
  assert torch.equal(dsts[2].t, want[2].t)
  one = [base.empty_like()]
  ops.input_builder(lab, ins, 35, one, [imgs[1]], 36)     # a single destination (discriminator skipped)
  assert torch.equal(one[0].t, want[1].t)
  gd = np.load(os.path.join(golden_dir, 'preprocess_cityscapes_crop.npz'))
  lab1 = torch.tensor(gd['label'].astype(np.float32))[None, None]
  lab1[lab1 == 255] = 35
  ins1 = torch.tensor(gd['instance'].astype(np.int64))[None, None]
  H, W = lab1.shape[2:]
  im1 = to_act(torch.zeros(1, 3, H, W), dtype)
  out = Act.empty(1, H, W, 39, dtype, DEV)
  ops.input_builder(lab1.to(DEV).contiguous(), ins1.to(DEV).contiguous(), 35, [out], [im1], 36)
  assert np.array_equal(to_nchw(out)[:, :36].numpy().astype(np.uint8), gd['input_label'])   # the reference's output


# ---- losses ------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', DTYPES)
def test_losses_fwd_bwd(dtype):
  g = G(11)
  a = quantize_like(torch.randn(2, 39, 9, 13, generator=g), dtype)
  b = quantize_like(torch.randn(2, 39, 9, 13, generator=g), dtype)
  A, Bm = to_act(a, dtype), to_act(b, dtype)
  slots = torch.zeros(4, device=DEV)
  one = torch.full((1,), 0.5, device=DEV)
  ops.l1_fwd(A, Bm, slots[0:1])
  ops.mse_fwd(A, Bm, slots[1:2])
  ar = a.clone().requires_grad_(True)
  F.l1_loss(ar, b).backward()
  tol = RTOL[dtype]
  assert_close(slots[0].cpu(), F.l1_loss(a, b), 1e-4, 'l1 fwd')
  assert_close(slots[1].cpu(), F.mse_loss(a, b), 1e-4, 'mse fwd')
  assert_close(to_nchw(ops.l1_bwd(A, Bm, one, 3.0)), 1.5 * ar.grad, tol, 'l1 bwd')
  ar.grad = None
  F.mse_loss(ar, b).backward()
  assert_close(to_nchw(ops.mse_bwd(A, Bm, one, 3.0)), 1.5 * ar.grad, tol, 'mse bwd')
  p = quantize_like(torch.randn(3, 1, 7, 11, generator=g), dtype)
  P = to_act(p, dtype)
  ops.mse_const_fwd(P, 1.0, slots[2:3])
  pr = p.clone().requires_grad_(True)
  F.mse_loss(pr, torch.ones_like(pr)).backward()
  assert_close(slots[2].cpu(), F.mse_loss(p, torch.ones_like(p)), 1e-4, 'mse_const fwd')
  dp = ops.mse_const_bwd(P, 1.0, one, 2.0)
  assert_close(to_nchw(dp), pr.grad, tol, 'mse_const bwd')
  assert (dp.t[..., 1:] == 0).all()


# ---- Adam --------------------------------------------------------------------------------------
def test_fused_adam_matches_torch():
  g = G(13)
  shapes = [(16, 8, 3, 3), (5,), (1030,), (3, 64, 7, 7)]
  ps = [torch.randn(s, generator=g) for s in shapes]
  ref = [p.clone().requires_grad_(True) for p in ps]
  mine = []
  for p in ps:
    q = p.clone().to(DEV)
    if q.dim() == 4:
      q = q.contiguous(memory_format=torch.channels_last)
    mine.append(torch.nn.Parameter(q))
  # fused bf16 copy of the updated parameter (the forward GEMM panel of plain conv layers); ragged tail too
  casts = {0: torch.zeros(mine[0].numel(), dtype=torch.bfloat16, device=DEV),
           2: torch.zeros(mine[2].numel(), dtype=torch.bfloat16, device=DEV)}
  for i, c in casts.items():
    mine[i]._jpdse_cast_out = c
  o_ref = torch.optim.Adam(ref, lr=2e-4, betas=(0.5, 0.999))
  o_hip = FusedAdam(mine, lr=2e-4, betas=(0.5, 0.999))
  for step in range(3):
    for r, m in zip(ref, mine):
      gr = torch.randn(r.shape, generator=g)
      r.grad = gr.clone()
      if m.grad is None:
        m.grad = torch.zeros_like(m, memory_format=torch.preserve_format)
      m.grad.copy_(gr)
    o_ref.step()
    o_hip.step()
  for r, m in zip(ref, mine):
    assert_close(m.detach().cpu(), r.detach(), 1e-6, 'adam param')
  for i, c in casts.items():
    m = mine[i].detach()
    flat = m.permute(0, 2, 3, 1).reshape(-1) if m.dim() == 4 else m.reshape(-1)     # memory (KRSC) order
    assert torch.equal(c, flat.bfloat16()), 'fused bf16 cast of the updated parameter'
    assert mine[i]._jpdse_cast_wver == mine[i]._jpdse_wver == 3
  sd = o_hip.state_dict()
  assert set(sd['state'][0].keys()) >= {'step', 'exp_avg', 'exp_avg_sq'} and float(sd['state'][0]['step']) == 3.0


# ---- full-size, oracle-free: the three convolution kernels of a layer are mutually adjoint -------------------------
# <conv(x), dy> = <x, dgrad(dy)> = <w, wgrad(x, dy)>.  The oracle finishes these shapes in minutes, not seconds, so at
# BASELINE.json's sizes (batch 4 @ 1024x512) parity is checked through this size-independent property: it ties the
# forward, data-gradient and weight-gradient kernels (three different tilings, loaders and epilogues) to one number.
FULL_SIZE_LAYERS = [
    # name,                  N, H,   W,    C,    K,    k, st, pad, mode
    ('resblock_1024',        4, 32,  64,   1024, 1024, 3, 1,  1,   PAD_REFLECT),
    ('g_first_7x7',          4, 512, 1024, 39,   64,   7, 1,  3,   PAD_REFLECT),
    ('g_down_64_128',        4, 512, 1024, 64,   128,  3, 2,  1,   PAD_ZERO),
    ('vgg_conv1_2',          4, 512, 1024, 64,   64,   3, 1,  1,   PAD_ZERO),
    ('d_layer0',             8, 512, 1024, 39,   64,   4, 2,  2,   PAD_ZERO),
    ('d_layer3',             8, 65,  129,  256,  512,  4, 1,  2,   PAD_ZERO),
]


@pytest.mark.parametrize('case', FULL_SIZE_LAYERS, ids=[c[0] for c in FULL_SIZE_LAYERS])
def test_full_size_adjointness_bf16(case):
  """Three seeds per layer; the bound of each layer is 3x its largest measured residual (hip_util.ADJ_BOUND).  Random x, dy
  make <y,dy> itself ~1e-4 of the scale (n ~ 1e8 terms of random sign): dropping one border ring at 1024x512 (0.2 % of the
  terms) moves the product by ~2e-7 of the scale per ring row -- the element-wise window checks of
  tests/test_hip_fullsize_windows.py are what sees a wrong row; this ties the three kernels of a layer to one number."""
  from hip_util import adjointness
  assert adjointness(*case, seeds=(0, 1, 2)), 'weight gradient not bit-reproducible'


def test_conv_entry_points_refuse_short_workspace_before_launching():
  """Every conv entry point checks the caller's workspace against jpdse_conv_workspace_size BEFORE anything is enqueued
  (JPDSE_EWORKSPACE): the reflect data gradient puts its ring-strip slabs, the split weight gradients their partial
  slabs there, and a short buffer would otherwise be a device memory fault."""
  import ctypes
  from jpdse_hip import lib, ConvDesc, last_error
  d = ops.conv_desc(BF16, 2, 8, 64, 64, 128, 3, 3, 1, 1, PAD_REFLECT)           # ring path + nine-tap weight gradient
  need = lib().jpdse_conv_workspace_size(ctypes.byref(d))
  assert need > 0
  layer = HipConv2d(64, 128, 3, 1, 1, PAD_REFLECT, apply_bias=False, dtype=BF16, device=DEV)
  fwd_pack, dgrad_pack = layer.packs()
  x = Act.empty(2, 8, 64, 64, BF16, DEV)
  x.t.zero_()
  dy = Act.empty(2, 8, 64, 128, BF16, DEV)
  dy.t.zero_()
  dx = x.empty_like()
  dx.t.fill_(7.0)
  dw = torch.full((128, 3, 3, 64), 7.0, device=DEV)
  ws = torch.empty(need, dtype=torch.uint8, device=DEV)
  s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
  p = lambda t: ctypes.c_void_p(t.data_ptr())
  short = need // 2
  assert lib().jpdse_conv_dgrad(ctypes.byref(d), p(dy.t), p(dgrad_pack), p(dx.t), p(ws), short, s) == -2
  assert 'workspace' in last_error()
  assert lib().jpdse_conv_wgrad(ctypes.byref(d), p(x.t), p(dy.t), p(dw), p(ws), short, s) == -2
  assert lib().jpdse_conv_fwd(ctypes.byref(d), p(x.t), p(fwd_pack), None, p(dy.t), p(ws), short, s) == -2
  assert lib().jpdse_conv_dgrad(ctypes.byref(d), p(dy.t), p(dgrad_pack), p(dx.t), None, need, s) == -2
  torch.cuda.synchronize()
  assert (dx.t == 7.0).all() and (dw == 7.0).all(), 'a refused call must not have launched anything'
  assert lib().jpdse_conv_dgrad(ctypes.byref(d), p(dy.t), p(dgrad_pack), p(dx.t), p(ws), need, s) == 0
  assert lib().jpdse_conv_wgrad(ctypes.byref(d), p(x.t), p(dy.t), p(dw), p(ws), need, s) == 0
  torch.cuda.synchronize()
  assert (dx.t == 0).all() and (dw == 0).all()
