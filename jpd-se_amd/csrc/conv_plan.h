// Host-side planner of the convolution family: descriptor validation, the plan of a layer (padded sizes, stride phases of
// the data gradient, panel sizes, workspace regions) and the padding launcher.  Runs without a GPU (jpdse_conv_plan_query).
// Part of conv_gemm.hip (one translation unit).
#pragma once

namespace jpdse {

// =========================================================================================
// host side: planning and launch
// =========================================================================================
static constexpr size_t kSlackBytes = 2048;  // readable, zeroed tail after every padded tensor

static inline int bke(int dtype) { return dtype == JPDSE_BF16 ? 32 : 16; }  // elements per 64-byte chunk
static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

static int validate(const jpdse_conv_desc* d) {
  JPDSE_REQUIRE(d != nullptr, "conv: null descriptor");
  JPDSE_REQUIRE(d->dtype == JPDSE_F32 || d->dtype == JPDSE_BF16, "conv: bad dtype %d", d->dtype);
  JPDSE_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->C > 0 && d->K > 0, "conv: non-positive shape");
  JPDSE_REQUIRE(d->R > 0 && d->S > 0 && d->R <= 16 && d->S <= 16, "conv: filter %dx%d unsupported", d->R, d->S);
  JPDSE_REQUIRE(d->stride == 1 || d->stride == 2, "conv: stride %d unsupported", d->stride);
  JPDSE_REQUIRE(d->R >= d->stride && d->S >= d->stride, "conv: filter smaller than stride");
  JPDSE_REQUIRE(d->pad >= 0, "conv: negative pad");
  JPDSE_REQUIRE(d->pad_mode == JPDSE_PAD_ZERO || d->pad_mode == JPDSE_PAD_REFLECT, "conv: bad pad mode");
  if (d->pad_mode == JPDSE_PAD_REFLECT) {
    JPDSE_REQUIRE(d->stride == 1, "conv: reflect padding requires stride 1");
    JPDSE_REQUIRE(d->pad < d->H && d->pad < d->W, "conv: reflect pad %d >= image dim", d->pad);
  }
  JPDSE_REQUIRE(d->H + 2 * d->pad >= d->R && d->W + 2 * d->pad >= d->S, "conv: image smaller than filter");
  return JPDSE_OK;
}

struct Phase {
  int qh, qw, Uh, Uw;
  int i0h, cnth, i0w, cntw;
  int Lk;
  size_t pack_off;  // bytes
};

struct ConvPlan {
  int ES, BKE;
  int Cs, Ks, Hp, Wp, OH, OW;
  int Lk_fwd;
  // dgrad
  int nph;
  Phase ph[4];
  int PT, PB, PL, PR;  // zero padding of dy
  int DH, DW;          // padded dy dims
  size_t dgrad_pack_bytes;
  size_t xpad_bytes, dypad_bytes, dxp_bytes;
  size_t splitk_off, splitk_bytes;   // fp32 partial slabs of the split-K fast path (behind the other regions)
  int toep, Lk_toep;                 // bf16 head (<= 8 output channels, stride 1): extra Toeplitz forward panel
  int thinf, KP_thin;                // bf16 stride-1 conv on a thin input: extra [R][K][KP] panel (thin_fwd.h)
  size_t thin_pack_off;
  size_t fwd_pack_plain_bytes, fwd_pack_bytes;
};

static void phase_axis(int st, int q, int Rf, int lo, int hi, int& U, int& i0, int& cnt) {
  U = (Rf - q + st - 1) / st;
  auto ceil_div = [](int a, int b) { return a >= 0 ? (a + b - 1) / b : -((-a) / b); };
  i0 = ceil_div(lo - q, st);
  const int i1 = ceil_div(hi - q, st);
  cnt = i1 - i0;
  if (cnt < 0) cnt = 0;
}

// Split-K factor of the fast 256x128 kernel for a GEMM of M rows, Ks output channels and k_tiles
// 64-wide K-tiles: only when the tiles alone would leave most of the 256 CUs idle.
JPDSE_SWITCH(int, g_splitk_enabled, 1);   // A/B switches (jpdse_debug_set_fast_path 6 / 5)
JPDSE_SWITCH(int, g_toep_enabled, 1);
JPDSE_SWITCH(int, g_thin_out_fast, 1);
static int splitk_for(int M, int Ks, int k_tiles) {
  if (!g_splitk_enabled) return 1;
  // narrow outputs (Ks <= 32: the 512 -> 1 PatchGAN map) only with long reductions
  if ((Ks <= 64 && !(Ks <= 32 && k_tiles >= 64 && g_thin_out_fast)) || k_tiles < 32 || M <= 0) return 1;
  const int bn = Ks > 64 ? 128 : (Ks > 32 ? 64 : 32);
  const long long tiles = (long long)((M + 255) / 256) * ((Ks + bn - 1) / bn);
  if (tiles > 128) return 1;
  int sp = (int)(256 / tiles);
  if (sp > k_tiles / 16) sp = k_tiles / 16;
  if (sp > 8) sp = 8;
  return sp < 2 ? 1 : sp;
}

// Split-K factor of the GENERIC kernel (gemm_fwd_kernel, fp32 layers): M rows, Ks output channels, `chunks` 64-byte K-chunks
// (16 fp32 elements each).  Only when the 128-row tiles alone leave most of the chip idle -- at 512x256 batch 1 (BASELINE
// config 2) the ResnetBlock GEMM is M = 512: 32 tiles on 256 CUs, 0.88 ms per launch; 16 splits: 512 blocks, three per CU.
JPDSE_SWITCH(int, g_generic_splitk, 1);   // A/B switch (jpdse_debug_set_fast_path 48 / 49)
// The factor is chosen from the rows of ONE image (M / N), not of the batch: the K ranges -- and with them every output value, bit
// for bit -- then do not depend on the batch size, so a 2-image step still equals the two 1-image steps exactly (the per-image
// independence test; data-parallel replicas with different local batch sizes agree).  Only when the whole batch would give more
// than 2048 blocks is the factor cut down (N > 4 at the sizes where it matters).
static int generic_splitk_for(int ES, int M, int Ks, int chunks, int N) {
  if (!g_generic_splitk || ES != 4 || M <= 0 || chunks < 32) return 1;
  if (N < 1) N = 1;
  const int bn = Ks > 64 ? 128 : (Ks > 32 ? 64 : 32), bm = Ks > 32 ? 128 : 256;
  const int nt = (Ks + bn - 1) / bn;
  const long long tiles_img = (long long)((M / N + bm - 1) / bm) * nt, tiles = (long long)((M + bm - 1) / bm) * nt;
  if (tiles_img >= 256) return 1;
  int sp = (int)((512 + tiles_img - 1) / tiles_img);
  if (sp > chunks / 16) sp = chunks / 16;         // >= 16 chunks (one two-level flush) per split
  if (sp > 32) sp = 32;
  while (sp > 1 && sp * tiles > 2048) sp = (sp + 1) / 2;
  return sp < 2 ? 1 : sp;
}

static void make_plan(const jpdse_conv_desc* d, ConvPlan* p) {
  p->ES = (int)esize(d->dtype);
  p->BKE = bke(d->dtype);
  p->Cs = cpad(d->C);
  p->Ks = cpad(d->K);
  p->Hp = d->H + 2 * d->pad;
  p->Wp = d->W + 2 * d->pad;
  p->OH = (p->Hp - d->R) / d->stride + 1;
  p->OW = (p->Wp - d->S) / d->stride + 1;
  p->Lk_fwd = round_up(d->S * p->Cs, p->BKE);
  const int st = d->stride;
  const bool refl = d->pad_mode == JPDSE_PAD_REFLECT;
  const int lo_h = refl ? 0 : d->pad, hi_h = refl ? p->Hp : d->pad + d->H;
  const int lo_w = refl ? 0 : d->pad, hi_w = refl ? p->Wp : d->pad + d->W;
  p->nph = 0;
  int min_h = 0, max_h = p->OH - 1, min_w = 0, max_w = p->OW - 1;
  size_t off = 0;
  for (int qh = 0; qh < st; ++qh)
    for (int qw = 0; qw < st; ++qw) {
      Phase& f = p->ph[p->nph++];
      f.qh = qh;
      f.qw = qw;
      phase_axis(st, qh, d->R, lo_h, hi_h, f.Uh, f.i0h, f.cnth);
      phase_axis(st, qw, d->S, lo_w, hi_w, f.Uw, f.i0w, f.cntw);
      f.Lk = round_up(f.Uw * p->Ks, p->BKE);
      f.pack_off = off;
      off += (size_t)p->Cs * f.Uh * f.Lk * p->ES;
      off = align_up(off, 256);
      if (f.cnth > 0 && f.cntw > 0) {
        min_h = min_h < f.i0h - (f.Uh - 1) ? min_h : f.i0h - (f.Uh - 1);
        max_h = max_h > f.i0h + f.cnth - 1 ? max_h : f.i0h + f.cnth - 1;
        min_w = min_w < f.i0w - (f.Uw - 1) ? min_w : f.i0w - (f.Uw - 1);
        max_w = max_w > f.i0w + f.cntw - 1 ? max_w : f.i0w + f.cntw - 1;
      }
    }
  p->dgrad_pack_bytes = off;
  p->PT = -min_h;
  p->PB = max_h - (p->OH - 1);
  p->PL = -min_w;
  p->PR = max_w - (p->OW - 1);
  p->DH = p->OH + p->PT + p->PB;
  p->DW = p->OW + p->PL + p->PR;
  p->xpad_bytes = align_up((size_t)d->N * p->Hp * p->Wp * p->Cs * p->ES + kSlackBytes, 256);
  p->dypad_bytes = align_up((size_t)d->N * p->DH * p->DW * p->Ks * p->ES + kSlackBytes, 256);
  p->dxp_bytes = refl ? align_up((size_t)d->N * p->Hp * p->Wp * p->Cs * p->ES, 256) : 0;
  p->fwd_pack_plain_bytes = align_up((size_t)p->Ks * d->R * p->Lk_fwd * p->ES, 256);
  p->toep = (p->ES == 2 && p->Ks == 8 && st == 1) ? 1 : 0;
  p->Lk_toep = p->toep ? round_up((d->S + 3) * p->Cs, p->BKE) : 0;
  p->fwd_pack_bytes = p->fwd_pack_plain_bytes + (p->toep ? align_up((size_t)32 * d->R * p->Lk_toep * p->ES, 256) : 0);
  p->thinf = (p->ES == 2 && p->Cs % 64 != 0 && p->Cs <= 48 && st <= 2 && (p->Ks == 32 || p->Ks == 64) && d->K == p->Ks) ? 1 : 0;
  p->KP_thin = p->thinf ? round_up(d->S * p->Cs, 16) + 8 : 0;
  p->thin_pack_off = p->fwd_pack_bytes;
  if (p->thinf) p->fwd_pack_bytes += align_up((size_t)d->R * p->Ks * p->KP_thin * 2, 256);
  if (p->toep) {
    // the Toeplitz rows of the last pixel group read (S+3)*Cs rounded up to a chunk: keep that inside the slack
    p->xpad_bytes = align_up(p->xpad_bytes + (size_t)p->BKE * p->ES, 256);
  }
  p->splitk_off = p->xpad_bytes > p->dypad_bytes + p->dxp_bytes ? p->xpad_bytes : p->dypad_bytes + p->dxp_bytes;
  p->splitk_bytes = 0;
  if (p->ES == 2) {
    const int sf = splitk_for(d->N * p->OH * p->OW, p->Ks, d->R * d->S * p->Cs / 64);
    const size_t fwd = sf > 1 ? (size_t)sf * d->N * p->OH * p->OW * p->Ks * 4 : 0;
    size_t dgr = 0;
    if (p->nph == 1 && p->ph[0].cnth > 0 && p->ph[0].cntw > 0) {
      const int Md = d->N * p->ph[0].cnth * p->ph[0].cntw;
      const int sd = splitk_for(Md, p->Cs, p->ph[0].Uh * p->ph[0].Uw * p->Ks / 64);
      dgr = sd > 1 ? (size_t)sd * Md * p->Cs * 4 : 0;
    }
    p->splitk_bytes = align_up(fwd > dgr ? fwd : dgr, 256);
  } else {
    // generic split-K (fp32): forward and every stride phase of the data gradient (run one after the other: the slabs are reused)
    const int Mf = d->N * p->OH * p->OW;
    size_t need = (size_t)generic_splitk_for(p->ES, Mf, p->Ks, d->R * (p->Lk_fwd / p->BKE), d->N) * Mf * p->Ks * 4;
    for (int i = 0; i < p->nph; ++i) {
      const Phase& f = p->ph[i];
      if (f.cnth <= 0 || f.cntw <= 0) continue;
      const int Md = d->N * f.cnth * f.cntw;
      const size_t b = (size_t)generic_splitk_for(p->ES, Md, p->Cs, f.Uh * (f.Lk / p->BKE), d->N) * Md * p->Cs * 4;
      need = need > b ? need : b;
    }
    p->splitk_bytes = align_up(need, 256);
  }
}

template <typename T>
static int launch_pad(const void* src, void* dst, int N, int H, int W, int Cs, int pt, int pb, int pl, int pr,
                      int mode, hipStream_t s) {
  const int Hp = H + pt + pb, Wp = W + pl + pr;
  const int VE = 16 / (int)sizeof(T);
  const long long total_vec = (long long)N * Hp * Wp * (Cs / VE);
  const long long slack_vec = kSlackBytes / 16;
  int tx_shift = 0;
  while ((1 << tx_shift) < Cs / VE && tx_shift < 8) ++tx_shift;
  int grid = N * Hp;
  if (grid > 4096) grid = 4096;
  // few rows (the deep layers at small resolutions: BASELINE config 2 at batch 1 has 18-row maps): split every row into column
  // segments until ~512 blocks exist, at least 4 pixels per thread row and segment
  const int TY = 256 >> tx_shift;
  int segs = 1;
  while (grid * segs < 512 && Wp / (segs * 2) >= 4 * TY && segs < 64) segs *= 2;
  hipLaunchKernelGGL((pad_kernel<T>), dim3(grid, segs), dim3(256), 0, s,
                     reinterpret_cast<const T*>(src), reinterpret_cast<T*>(dst), N, H, W, Cs, pt, pl, Hp, Wp,
                     mode, tx_shift, total_vec, slack_vec);
  return check_launch("pad_kernel");
}

}  // namespace jpdse
