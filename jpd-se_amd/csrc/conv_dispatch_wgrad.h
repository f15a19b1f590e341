// Weight-gradient launchers and dispatch of the convolution family (all split reductions go through fp32 slabs summed in a
// fixed order: no atomics).  Part of conv_gemm.hip (one translation unit).
#pragma once

namespace jpdse {

// (tile, split) partition of the fast weight-gradient kernel
template <int BM, int BN>
static void fast_wgrad_partition(FastWgArgs* a, int lds) {
  const int blocks_per_cu = lds <= 80 * 1024 ? 2 : 1;
  a->chunks_total = (a->M + 63) / 64;
  const int tiles = ((a->Ks + BM - 1) / BM) * (a->run_mode ? a->R : a->R * a->S) *
                    (((a->run_mode ? a->run_len : a->Cs) + BN - 1) / BN);
  // Split count by a small cost model instead of "fill the chip once" (round 3): rounds of resident blocks x (chunks per block
  // + a fixed per-block cost of ~6 chunk times for prologue, epilogue and the 256 KB tile store) + the slab reduction a
  // split costs (~4 + 2 per slab).  "Fill once" chose 2 splits for 144 tiles -- 288 blocks = two rounds, the second with 32
  // blocks -- plus a 113 MB reduction, where one unsplit round is shorter (LocalEnhancer trunk: M = 2048 pixels).
  const int max_splits = a->chunks_total / 8 > 0 ? a->chunks_total / 8 : 1;      // >= 8 chunks per block
  const long long slots = 256LL * blocks_per_cu;
  int splits = 1;
  long long best = -1;
  for (int sp = 1; sp <= max_splits && sp <= 64; ++sp) {
    const long long rounds = ((long long)tiles * sp + slots - 1) / slots;
    const long long cost = rounds * ((a->chunks_total + sp - 1) / sp + 6) + (sp > 1 ? 4 + 2 * sp : 0);
    if (best < 0 || cost < best) { best = cost; splits = sp; }
  }
  a->chunks_per_split = (a->chunks_total + splits - 1) / splits;
  a->splits = (a->chunks_total + a->chunks_per_split - 1) / a->chunks_per_split;
  a->slab_stride = ((long long)a->K * a->R * a->S * a->C + 3) / 4 * 4;
}

template <int WM, int WN, int TM, int TN, int ABL = 0>
static int launch_wgrad_fast_cfg(FastWgArgs a, float* slabs, size_t* slab_bytes_out, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int lds = 2 * 64 * 2 * (BM + BN);
  fast_wgrad_partition<BM, BN>(&a, lds);
  if (slab_bytes_out != nullptr) {        // workspace query only
    *slab_bytes_out = a.splits > 1 ? (size_t)a.splits * a.slab_stride * sizeof(float) : 0;
    return JPDSE_OK;
  }
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_fast_kernel<WM, WN, TM, TN, ABL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "wgrad_fast: hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  const int tiles = ((a.Ks + BM - 1) / BM) * (a.run_mode ? a.R : a.R * a.S) *
                    (((a.run_mode ? a.run_len : a.Cs) + BN - 1) / BN);
  a.partial = slabs;
  hipLaunchKernelGGL((wgrad_fast_kernel<WM, WN, TM, TN, ABL>), dim3(tiles * a.splits), dim3(64 * WM * WN), lds, s, a);
  if (int rc = check_launch("wgrad_fast_kernel")) return rc;
  return a.splits > 1 ? launch_slab_reduce(slabs, a.DW, (long long)a.K * a.R * a.S * a.C, a.slab_stride, a.splits, s) : JPDSE_OK;
}

// slab_bytes_out != nullptr: report the slab bytes the launch would need instead of launching (workspace query)
static int launch_wgrad_fast(const FastWgArgs& a, float* slabs, hipStream_t s, size_t* slab_bytes_out = nullptr) {
  const int cols = a.run_mode ? a.run_len : a.Cs;
  const bool m2 = a.Ks >= 128, n2 = cols >= 128;
  // 256 x 256 tiles, 8 waves, one block per CU.  (Round 3 tried 128 x 128 tiles, two blocks per CU, for the short reductions
  // of the LocalEnhancer trunk -- 144 large tiles cover 56 % of the chip -- and measured it 0.45 ms per step SLOWER.)
  if (a.Ks >= 256 && cols >= 256 && a.Ks % 256 == 0 && cols % 256 == 0 && !a.run_mode)
    return launch_wgrad_fast_cfg<2, 4, 4, 2>(a, slabs, slab_bytes_out, s);
  if (m2 && n2) return launch_wgrad_fast_cfg<2, 2, 2, 2>(a, slabs, slab_bytes_out, s);
  if (m2) return launch_wgrad_fast_cfg<2, 2, 2, 1>(a, slabs, slab_bytes_out, s);
  if (n2) return launch_wgrad_fast_cfg<2, 2, 1, 2>(a, slabs, slab_bytes_out, s);
  return launch_wgrad_fast_cfg<2, 2, 1, 1>(a, slabs, slab_bytes_out, s);
}

template <int TM, int NW, int NT, int RR, int WM, int PITCH>
static int launch_wgrad_thin_pitch(ThinWgArgs a, hipStream_t s);

template <int TM, int NW, int NT, int RR = 1, int WM = 1>
static int launch_wgrad_thin_cfg(const ThinWgArgs& a, hipStream_t s) {
  switch (a.st * a.Cs) {      // the pitches of the hot path as compile-time constants
    case 40: return launch_wgrad_thin_pitch<TM, NW, NT, RR, WM, 40>(a, s);   // 39-channel inputs, stride 1
    case 80: return launch_wgrad_thin_pitch<TM, NW, NT, RR, WM, 80>(a, s);   // 39-channel inputs, stride 2
    case 8: return launch_wgrad_thin_pitch<TM, NW, NT, RR, WM, 8>(a, s);     // heads (dy run operand)
    default: return launch_wgrad_thin_pitch<TM, NW, NT, RR, WM, 0>(a, s);
  }
}

static int thin_wgrad_ranges(const ThinWgArgs& a, int RR, int* strips_per_block) {
  const int chunks_per_row = (a.OW + 63) / 64;
  const int strips_total = a.N * a.OH * chunks_per_row;
  const int row_groups = (a.R + RR - 1) / RR;
  int P = 1024 / row_groups;               // ~4 blocks per CU over the filter-row groups
  if (P < 1) P = 1;
  if (P > strips_total) P = strips_total;
  const int spb = (strips_total + P - 1) / P;
  if (strips_per_block) *strips_per_block = spb;
  return (strips_total + spb - 1) / spb;
}
static size_t thin_wgrad_slab_bytes(const ThinWgArgs& a, int RR) {
  const long long n = ((long long)a.K * a.R * a.S * a.C + 3) / 4 * 4;
  return (size_t)thin_wgrad_ranges(a, RR, nullptr) * n * sizeof(float);
}

template <int TM, int NW, int NT, int RR, int WM, int PITCH>
static int launch_wgrad_thin_pitch(ThinWgArgs a, hipStream_t s) {
  const int pitch = a.st * a.Cs;
  a.x_units = (126 * pitch + 64 * NW * NT + 1023) / 1024;
  const int lds = 2 * (TM * WM * 64 * 64 + RR * a.x_units * 1024);
  if (lds > 64 * 1024) return set_error(JPDSE_ELAUNCH, "wgrad_thin: strip of %d B does not fit", lds);
  a.chunks_per_row = (a.OW + 63) / 64;
  a.strips_total = a.N * a.OH * a.chunks_per_row;
  a.row_groups = (a.R + RR - 1) / RR;
  a.ranges = thin_wgrad_ranges(a, RR, &a.strips_per_block);
  a.slab_stride = ((long long)a.K * a.R * a.S * a.C + 3) / 4 * 4;
  if (a.partial == nullptr) return set_error(JPDSE_EWORKSPACE, "wgrad_thin: no slab workspace");
  // block -> (pixel range, filter-row group): the row groups of ONE pixel range read the same dy strips and nearly the
  // same input rows; they get consecutive slots of one XCD (blocks b, b + 8, ... share an XCD: observed dispatch,
  // speed only), so its L2 serves them -- as a (ranges, row_groups) grid they ran far apart in time and every row
  // group re-fetched x and dy from beyond L2 (7x the algorithmic bytes on the first 7x7 conv)
  const int blocks = ((a.ranges + 7) / 8) * 8 * a.row_groups;
  if constexpr (RR > 1) {      // heads: roles swapped, transposed output
    hipLaunchKernelGGL((wgrad_thin_kernel<TM, WM, NW, NT, RR, PITCH, true>), dim3(blocks), dim3(64 * WM * NW), lds, s, a);
  } else {
    hipLaunchKernelGGL((wgrad_thin_kernel<TM, WM, NW, NT, RR, PITCH, false>), dim3(blocks), dim3(64 * WM * NW), lds, s, a);
  }
  if (int rc = check_launch("wgrad_thin_kernel")) return rc;
  return launch_slab_reduce(a.partial, a.DW, (long long)a.K * a.R * a.S * a.C, a.slab_stride, a.ranges, s);
}

static bool wgrad_thin_ok(const jpdse_conv_desc* d, const ConvPlan& p) {
  const int n_tiles = (d->S * p.Cs + 31) / 32;
  return g_fast_enabled && p.Cs % 64 != 0 && (p.Ks == 32 || p.Ks == 64) && n_tiles <= 12 &&
         (126 * d->stride * p.Cs + 64 * 12) <= 20 * 1024;
}

static int launch_wgrad_thin(const ThinWgArgs& a, hipStream_t s) {
  const int n_tiles = (a.S * a.Cs + 31) / 32;
  if (a.Ks == 64) {
    if (n_tiles <= 6) return launch_wgrad_thin_cfg<2, 2, 3>(a, s);
    if (n_tiles <= 9) return launch_wgrad_thin_cfg<2, 3, 3>(a, s);
    return launch_wgrad_thin_cfg<2, 4, 3>(a, s);
  }
  if (n_tiles <= 6) return launch_wgrad_thin_cfg<1, 2, 3>(a, s);
  if (n_tiles <= 9) return launch_wgrad_thin_cfg<1, 3, 3>(a, s);
  return launch_wgrad_thin_cfg<1, 4, 3>(a, s);
}

// ---- all-taps weight gradient of the narrow high-resolution layers (wgrad_taps.h) -----------------
// config id: 0 none; 1: 3x3 s2 K%128 C%64; 2: 3x3 s1 K%64 C%64; 3: 4x4 s2 K%128 C%64; 4: 3x3 s2 K%256 C%128
JPDSE_SWITCH(int, g_wgrad_taps_enabled, 1);
JPDSE_SWITCH(int, g_wgrad_taps_abl, 0);           // 200 + bits: timing-only ablations of the all-taps loop (developer build)
JPDSE_SWITCH(int, g_wgrad_taps_xcd, 0);           // 58: XCD co-location of the tiles of a pixel range (A/B; measured 0-13 % slower)
static int wgrad_taps_cfg(const jpdse_conv_desc* d, const ConvPlan& p) {
  if (!g_fast_enabled || !g_wgrad_taps_enabled || p.ES != 2 || d->R != d->S) return 0;
  if ((long long)d->N * p.OH * ((p.OW + 63) / 64) < 32) return 0;
  if ((long long)d->N * p.OH * p.OW * p.Ks >= (1LL << 31) || (long long)d->N * d->H * d->W * p.Cs >= (1LL << 31)) return 0;
  // 3x3 stride 2 with wide outputs, any width: the down-sampling convs 128 -> 256 ... 512 -> 1024 and, with the roles of
  // x and dy swapped by the caller, the ConvTranspose2d layers 1024 -> 512 ... 128 -> 64 (round 1 sent the wide ones to
  // the per-tap kernel, whose stream-K partial tiles met in fp32 atomics)
  if (d->R == 3 && d->stride == 2 && p.Ks % 256 == 0 && p.Cs % 64 == 0) return 4;      // (round 3: the all-nine-taps config 1 on these wide layers measured -3 ... -11 %, +5 % only at 512 -> 1024)
  if (p.Ks > 256 || p.Cs > 128) return 0;
  if (d->R == 3 && d->stride == 2 && p.Ks % 128 == 0 && p.Cs % 64 == 0) return 1;
  if (d->R == 3 && d->stride == 1 && p.Ks % 64 == 0 && p.Ks <= 128 && p.Cs % 64 == 0) return 2;
  if (d->R == 4 && d->stride == 2 && p.Ks % 128 == 0 && p.Cs % 64 == 0) return 3;
  return 0;
}

struct TapsGeom { int BM, BN, T, NROW, lds, blocks_per_cu, bkp; };
// LDS of one launch: stages x (dy tile + patch units + one scratch unit), wgrad_taps.h
static constexpr int taps_lds(int BM, int S, int NROW, int ST, int bkp, int stages) {
  return stages * (bkp * BM * 2 + ((NROW * ((bkp - 1) * ST + S) + 7) / 8 + 1) * 1024);
}
static TapsGeom taps_geom(int cfg) {
  switch (cfg) {
    // (32-pixel chunks through a four-stage ring, two chunks in flight at every wait, measured 5-20 % SLOWER on the stride-2 layers than
    // 64-pixel chunks / two stages, and three stages of 64-pixel chunks measured equal, profiles/r04_wgrad_taps_ab.txt: the fill
    // latency is already hidden -- what the DMA costs is the loader's instructions, wgrad_taps.h)
    case 1: return {128, 64, 9, 3, taps_lds(128, 3, 3, 2, 64, 2), 1, 64};
    case 2: return {64, 64, 9, 3, taps_lds(64, 3, 3, 1, 64, 2), 2, 64};
    case 3: return {128, 64, 8, 2, taps_lds(128, 4, 2, 2, 64, 2), 1, 64};
    default: return {256, 64, 3, 1, taps_lds(256, 3, 1, 2, 64, 2), 1, 64};
  }
}

static void taps_partition(const jpdse_conv_desc* d, const ConvPlan& p, int cfg, TapsWgArgs* a) {
  const TapsGeom g = taps_geom(cfg);
  a->chunks_per_row = (p.OW + g.bkp - 1) / g.bkp;
  a->chunks_total = d->N * p.OH * a->chunks_per_row;
  a->k_tiles = p.Ks / g.BM;
  a->r_groups = (d->R + g.NROW - 1) / g.NROW;
  a->c_tiles = p.Cs / g.BN;
  const int tiles = a->k_tiles * a->r_groups * a->c_tiles;
  a->tiles = tiles;
  int bpt = 256 * g.blocks_per_cu / tiles;
  if (bpt < 1) bpt = 1;
  // XCD co-location (round 4 experiment, developer mode 58, NOT the default; MI355X_MICROARCH.md "XCD placement", block b runs on XCD
  // b % 8).  The tiles of one pixel range read the same dy chunks (and, per input-channel tile, the same input rows); dealt linearly they
  // sit on different XCDs and every one of them pulls its own copy through the fabric (memory-side fetch 450-540 MB per launch for
  // ~100 MB of operands, profiles/r04_hbm_traffic.txt).  A GROUP = all tiles of a pixel range (<= 32 of them), else the tiles of one
  // (pixel range, k tile) -- the readers of one dy tile; an XCD gets whole groups, as many as its 32 CUs hold.  Measured
  // (profiles/r04_wgrad_taps_ab.txt): 0-13 % SLOWER -- whole groups per XCD leave CUs idle (24 tiles: 192 blocks instead of 240) and
  // the loop is not bound by where its operands come from.
  a->xg_gs = 0;
  a->xg_gpx = 0;
  const int cus = 32 * g.blocks_per_cu;
  const int gs = tiles <= cus ? tiles : a->r_groups * a->c_tiles;
  if (g_wgrad_taps_xcd && tiles > 1 && gs <= cus && tiles % gs == 0) {
    const int parts = tiles / gs;                       // groups per pixel range
    const int gpx = cus / gs;                           // groups one XCD holds at a time
    const int ranges = 8 * gpx / parts;                 // pixel ranges: 8 * gpx groups in all
    if (ranges >= 1 && (8 * gpx) % parts == 0 && ranges <= a->chunks_total) {
      bpt = ranges;
      a->xg_gs = gs;
      a->xg_gpx = gpx;
    }
  }
  if (bpt > a->chunks_total) bpt = a->chunks_total;
  a->chunks_per_block = (a->chunks_total + bpt - 1) / bpt;
  a->blocks_per_tile = (a->chunks_total + a->chunks_per_block - 1) / a->chunks_per_block;
  if (a->xg_gs) {
    const int groups = a->blocks_per_tile * (tiles / a->xg_gs);
    if (groups % 8 != 0) a->xg_gs = 0;                  // ragged range count: linear order
    else a->xg_gpx = groups / 8;
  }
}

static size_t wgrad_taps_ws_bytes(const jpdse_conv_desc* d, const ConvPlan& p) {
  const int cfg = wgrad_taps_cfg(d, p);
  if (!cfg) return 0;
  TapsWgArgs a = {};
  taps_partition(d, p, cfg, &a);
  const TapsGeom g = taps_geom(cfg);
  return (size_t)a.k_tiles * a.r_groups * a.c_tiles * a.blocks_per_tile * g.T * g.BM * g.BN * sizeof(float);
}

template <int TMW, int WM, int WN, int S, int NROW, int ST, int BKP, int NSTG>
static int launch_wgrad_taps_cfg(const TapsWgArgs& a, int lds, hipStream_t s) {
  constexpr int BM = WM * TMW * 32, BN = WN * 32;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_taps_kernel<TMW, WM, WN, S, NROW, ST, BKP, NSTG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "wgrad_taps: hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  const int blocks = a.k_tiles * a.r_groups * a.c_tiles * a.blocks_per_tile;
  hipLaunchKernelGGL((wgrad_taps_kernel<TMW, WM, WN, S, NROW, ST, BKP, NSTG>), dim3(blocks), dim3(64 * WM * WN), lds, s, a);
  if (int rc = check_launch("wgrad_taps_kernel")) return rc;
  const long long total = (long long)a.K * a.R * a.S * a.C;
  if (a.C % 4 == 0 && (reinterpret_cast<uintptr_t>(a.DW) & 15) == 0)
    hipLaunchKernelGGL((wgrad_taps_reduce_kernel<BM, BN, S, NROW, 4>), dim3((unsigned)((total / 4 + 63) / 64)), dim3(256), 0, s,
                       a, total / 4);
  else
    hipLaunchKernelGGL((wgrad_taps_reduce_kernel<BM, BN, S, NROW, 1>), dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s,
                       a, total);
  return check_launch("wgrad_taps_reduce_kernel");
}

static int launch_wgrad_taps(const jpdse_conv_desc* d, const ConvPlan& p, int cfg, const void* x, const void* dy,
                             float* dw, void* ws, hipStream_t s) {
  TapsWgArgs a = {};
  a.X = reinterpret_cast<const bf16_t*>(x);
  a.DY = reinterpret_cast<const bf16_t*>(dy);
  a.partial = reinterpret_cast<float*>(ws);
  a.DW = dw;
  a.N = d->N;
  a.IH = d->H;
  a.IW = d->W;
  a.OH = p.OH;
  a.OW = p.OW;
  a.Cs = p.Cs;
  a.C = d->C;
  a.Ks = p.Ks;
  a.K = d->K;
  a.R = d->R;
  a.S = d->S;
  a.pad = d->pad;
  a.reflect = d->pad_mode == JPDSE_PAD_REFLECT;
  taps_partition(d, p, cfg, &a);
  a.abl = g_wgrad_taps_abl;
  const int lds = taps_geom(cfg).lds;
  switch (cfg) {
    case 1: return launch_wgrad_taps_cfg<1, 4, 2, 3, 3, 2, 64, 2>(a, lds, s);
    case 2: return launch_wgrad_taps_cfg<1, 2, 2, 3, 3, 1, 64, 2>(a, lds, s);
    case 3: return launch_wgrad_taps_cfg<1, 4, 2, 4, 2, 2, 64, 2>(a, lds, s);
    default: return launch_wgrad_taps_cfg<2, 4, 2, 3, 1, 2, 64, 2>(a, lds, s);
  }
}

// ---- all-nine-taps weight gradient of the wide 3x3 stride-1 layers (wgrad_nine.h): no atomics, no partial tiles ----
JPDSE_SWITCH(int, g_wgrad_nine_enabled, 1);
JPDSE_SWITCH(int, g_wgrad_nine32_enabled, 1);     // 55: 32-pixel-wide images on the per-tap kernel, as before round 4 (A/B)
// images 32 pixels wide (the LocalEnhancer's 1024-channel trunk at 16 x 32): the row-pair form of the kernel (wgrad_nine.h, W32)
static bool wgrad_nine32_shape(const jpdse_conv_desc* d, const ConvPlan& p) {
  return g_wgrad_nine32_enabled && d->W == 32 && p.OW == 32 && d->H % 2 == 0 && d->H >= 4 && d->pad_mode == JPDSE_PAD_REFLECT &&
         p.Ks >= 256 && p.Cs >= 256;         // wide layers only: 64 k x 64 c tiles must fill the chip
}
static bool wgrad_nine_ok(const jpdse_conv_desc* d, const ConvPlan& p) {
  return g_fast_enabled && g_wgrad_nine_enabled && p.ES == 2 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 &&
         (p.OW % 64 == 0 || wgrad_nine32_shape(d, p)) && p.Ks % 64 == 0 && p.Cs % 64 == 0 && d->H >= 2 && d->W >= 8 &&
         (long long)d->N * d->H * d->W * (p.Ks > p.Cs ? p.Ks : p.Cs) < (1LL << 31);
}

static void nine_partition(const jpdse_conv_desc* d, const ConvPlan& p, NineWgArgs* a) {
  const bool w32 = p.OW % 64 != 0;           // wgrad_nine_ok: then the row-pair form (chunk = two image rows)
  a->strips = w32 ? 1 : d->W / 64;
  a->chunks_total = w32 ? d->N * (d->H / 2) : d->N * a->strips * d->H;
  a->k_tiles = p.Ks / 64;
  a->c_tiles = p.Cs / 64;
  const int tiles = a->k_tiles * a->c_tiles;
  int splits = 1;
  if (tiles < 192) {                         // fewer tiles than CUs: cut the pixel range (fp32 slabs + fixed-order reduce)
    splits = 256 / tiles;
    const int max_splits = a->chunks_total / 4 > 0 ? a->chunks_total / 4 : 1;    // >= 4 chunks per block
    if (splits > max_splits) splits = max_splits;
    if (splits > 32) splits = 32;
    if (splits < 1) splits = 1;
  }
  a->chunks_per_split = (a->chunks_total + splits - 1) / splits;
  a->splits = (a->chunks_total + a->chunks_per_split - 1) / a->chunks_per_split;
  a->xcd_map = (a->k_tiles % 4 == 0 && a->c_tiles % 8 == 0 && tiles % 256 == 0) ? 1 : 0;
}

static size_t wgrad_nine_ws_bytes(const jpdse_conv_desc* d, const ConvPlan& p) {
  if (!wgrad_nine_ok(d, p)) return 0;
  NineWgArgs a = {};
  nine_partition(d, p, &a);
  return a.splits > 1 ? (size_t)a.splits * d->K * 9 * d->C * sizeof(float) : 0;
}

template <bool REFLECT, int SCHED, bool W32 = false>
static int launch_wgrad_nine_cfg(const NineWgArgs& a, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_nine_kernel<REFLECT, SCHED, 0, W32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kNineLds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "wgrad_nine: hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  const int blocks = a.k_tiles * a.c_tiles * a.splits;
  const int pslot = (a.Ks == g_prof.Ks && 9LL * a.Cs == g_prof.kdim) ? prof_begin(s) : -1;
  if (W32 && (a.W != 32 || (a.H & 1) || a.chunks_total != a.N * (a.H / 2)))
    return set_error(JPDSE_EINVAL, "wgrad_nine: the row-pair form needs 32-pixel-wide images with an even number of rows");
  hipLaunchKernelGGL((wgrad_nine_kernel<REFLECT, SCHED, 0, W32>), dim3(blocks), dim3(512), kNineLds, s, a);
  if (int rc = check_launch("wgrad_nine_kernel")) return rc;
  if (a.splits > 1) {
    const long long n4 = (long long)a.K * 9 * a.C / 4;
    hipLaunchKernelGGL(wgrad_nine_reduce_kernel, dim3(ew_blocks(n4)), dim3(256), 0, s, a.partial, a.DW, n4, a.splits);
    if (int rc = check_launch("wgrad_nine_reduce_kernel")) return rc;
  }
  prof_end(pslot, 2, 2.0 * (double)a.N * a.H * a.W * (double)a.Ks * 9.0 * (double)a.Cs, s);
  return JPDSE_OK;
}

static int launch_wgrad_nine(const jpdse_conv_desc* d, const ConvPlan& p, const void* x, const void* dy, float* dw,
                             void* ws, hipStream_t s) {
  NineWgArgs a = {};
  a.X = reinterpret_cast<const bf16_t*>(x);
  a.DY = reinterpret_cast<const bf16_t*>(dy);
  a.DW = dw;
  a.partial = reinterpret_cast<float*>(ws);
  a.N = d->N;
  a.H = d->H;
  a.W = d->W;
  a.Cs = p.Cs;
  a.C = d->C;
  a.Ks = p.Ks;
  a.K = d->K;
  nine_partition(d, p, &a);
  if (a.splits > 1 && ((long long)d->K * 9 * d->C) % 4 != 0)
    return set_error(JPDSE_EINVAL, "wgrad_nine: K*9*C = %lld is not a multiple of 4", (long long)d->K * 9 * d->C);
  if (p.OW % 64 != 0) return launch_wgrad_nine_cfg<true, 3, true>(a, s);          // wgrad_nine32_shape: reflect, W = 32
  if (d->pad_mode != JPDSE_PAD_REFLECT) return launch_wgrad_nine_cfg<false, 3>(a, s);
  return launch_wgrad_nine_cfg<true, 3>(a, s);
}

// heads with <= 8 output channels on a 32- / 64-channel input, stride 1 (64->3, 32->3 7x7)
static bool wgrad_head_ok(const jpdse_conv_desc* d, const ConvPlan& p) {
  return g_fast_enabled && p.Ks == 8 && d->stride == 1 && (p.Cs == 32 || p.Cs == 64) && d->S * 8 <= 64 && d->R <= 7 &&
         d->R == d->S;
}

static int launch_wgrad_head(const ThinWgArgs& a, hipStream_t s) {
  // 2 waves x 1 run tile (S*8 <= 64 columns), all R <= 7 filter rows per block
  return a.Ks == 64 ? launch_wgrad_thin_cfg<1, 2, 1, 7, 2>(a, s) : launch_wgrad_thin_cfg<1, 2, 1, 7>(a, s);
}

// Workspace layout of the weight-gradient paths: [0, front) = padded copy of x (+ tap-expanded dy of the narrow-output
// layers), [front, ...) = the fp32 slabs of the split reductions (slab_reduce_kernel).  The all-taps kernels (wgrad_taps,
// wgrad_nine) make no copies and put their slabs at offset 0.
static size_t wgrad_front_bytes(const jpdse_conv_desc* d, const ConvPlan& p) {
  const size_t kexp_s = (size_t)round_up(d->K * d->R * d->S, 64);
  size_t dz = (p.Ks == 8 && kexp_s <= 256) ? align_up((size_t)d->N * p.Hp * p.Wp * kexp_s * 2, 256) : 0;
  if (p.Ks == 8 && d->stride == 1) {      // zero-padded dy of the head weight gradient (wgrad_thin.h, transposed)
    const size_t dyp = align_up((size_t)d->N * (p.OH + 2 * (d->R - 1)) * (p.OW + 2 * (d->S - 1)) * 8 * 2 + kSlackBytes, 256);
    dz = dz > dyp ? dz : dyp;
  }
  return align_up(p.xpad_bytes + dz, 256);
}

template <typename T>
static int conv_wgrad_t(const jpdse_conv_desc* d, const ConvPlan& p, const void* x, const void* dy, float* dw,
                        void* ws, hipStream_t s, size_t* slab_bytes_out = nullptr) {
  // slab_bytes_out != nullptr: dry run for jpdse_conv_workspace_size -- follows the dispatch below and reports the slab
  // bytes of the path that would run, launching nothing
  float* const slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + wgrad_front_bytes(d, p));
  const bool dry = slab_bytes_out != nullptr;
  if (dry) *slab_bytes_out = 0;
  if constexpr (sizeof(T) == 2) {
    if (thin1_shape_ok(d, p.Cs, p.Ks)) {
      // one output channel (PatchGAN 512 -> 1): x read once by an FMA kernel, no copies (thin_out1.h)
      if (dry) {
        *slab_bytes_out = thin1_wgrad_slab_bytes(d->N, d->H, d->W, p.Cs);
        return JPDSE_OK;
      }
      Thin1WgradArgs t = {};
      t.X = reinterpret_cast<const bf16_t*>(x);
      t.DY = reinterpret_cast<const bf16_t*>(dy);
      t.partial = slabs;
      t.N = d->N;
      t.H = d->H;
      t.W = d->W;
      t.Cs = p.Cs;
      t.OH = p.OH;
      t.OW = p.OW;
      t.pad = d->pad;
      return launch_thin1_wgrad(t, dw, s);
    }
    const int kexp = d->K * d->R * d->S, kexp_s = round_up(kexp, 64);
    if (g_fast_enabled && !wgrad_head_ok(d, p) && p.Ks == 8 && d->stride == 1 && p.Cs % 64 == 0 && kexp_s <= 256) {
      // few output channels: dense 1x1 weight gradient over the tap-expanded dy (see expand_dy_taps_kernel)
      bf16_t* dz = reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(ws) + p.xpad_bytes);
      FastWgArgs f = {};
      f.X = reinterpret_cast<const bf16_t*>(ws);
      f.DY = dz;
      f.DW = dw;
      f.M = d->N * p.Hp * p.Wp;
      f.OH = p.Hp;
      f.OW = p.Wp;
      f.IH = p.Hp;
      f.IW = p.Wp;
      f.Cs = p.Cs;
      f.C = d->C;
      f.Ks = kexp_s;
      f.K = kexp;
      f.R = f.S = 1;
      f.sy = f.sx = 1;
      f.py = f.px = 0;
      f.reflect = 0;
      if (dry) return launch_wgrad_fast(f, nullptr, s, slab_bytes_out);
      if (int rc = launch_pad<T>(x, ws, d->N, d->H, d->W, p.Cs, d->pad, d->pad, d->pad, d->pad, d->pad_mode, s))
        return rc;
      const long long tv = (long long)d->N * p.Hp * p.Wp * (kexp_s / 8);
      hipLaunchKernelGGL(expand_dy_taps_kernel, dim3(ew_blocks(tv)), dim3(256), 0, s,
                         reinterpret_cast<const bf16_t*>(dy), dz, p.OH, p.OW, p.Ks, d->K, d->R, d->S, p.Hp, p.Wp,
                         kexp_s, tv);
      if (int rc = check_launch("expand_dy_taps_kernel")) return rc;
      return launch_wgrad_fast(f, slabs, s);
    }
    if (const int tcfg = wgrad_taps_cfg(d, p)) return dry ? JPDSE_OK : launch_wgrad_taps(d, p, tcfg, x, dy, dw, ws, s);
    if (wgrad_nine_ok(d, p) && ((long long)d->K * 9 * d->C) % 4 == 0)
      return dry ? JPDSE_OK : launch_wgrad_nine(d, p, x, dy, dw, ws, s);
    if (wgrad_head_ok(d, p)) {
      // roles swapped (see wgrad_thin.h): A = padded input, run operand = dy zero-padded by (R-1, S-1); both
      // paddings are resolved by the loader
      ThinWgArgs t = {};
      t.XP = reinterpret_cast<const bf16_t*>(dy);
      t.DY = reinterpret_cast<const bf16_t*>(x);
      t.DW = dw;
      t.N = d->N;
      t.OH = p.Hp;
      t.OW = p.Wp;
      t.Hp = p.OH + 2 * (d->R - 1);
      t.Wp = p.OW + 2 * (d->S - 1);
      t.Cs = 8;
      t.C = d->K;
      t.Ks = p.Cs;
      t.K = d->C;
      t.R = d->R;
      t.S = d->S;
      t.st = 1;
      t.unpadded = 1;
      t.RH = p.OH;
      t.RW = p.OW;
      t.r_pad = d->R - 1;            // square filters on this path (R == S checked by wgrad_head_ok)
      t.r_reflect = 0;
      t.AH = d->H;
      t.AW = d->W;
      t.a_pad = d->pad;
      t.a_reflect = d->pad_mode == JPDSE_PAD_REFLECT;
      t.transposed = 1;
      t.partial = slabs;
      if (dry) {
        *slab_bytes_out = thin_wgrad_slab_bytes(t, 7);
        return JPDSE_OK;
      }
      return launch_wgrad_head(t, s);
    }
    if (wgrad_thin_ok(d, p)) {
      // thin inputs (40-channel network inputs): input strips staged once (wgrad_thin.h)
      ThinWgArgs t = {};
      t.XP = reinterpret_cast<const bf16_t*>(x);
      t.DY = reinterpret_cast<const bf16_t*>(dy);
      t.DW = dw;
      t.N = d->N;
      t.OH = p.OH;
      t.OW = p.OW;
      t.Hp = p.Hp;
      t.Wp = p.Wp;
      t.Cs = p.Cs;
      t.C = d->C;
      t.Ks = p.Ks;
      t.K = d->K;
      t.R = d->R;
      t.S = d->S;
      t.st = d->stride;
      t.partial = slabs;
      if (dry) {
        *slab_bytes_out = thin_wgrad_slab_bytes(t, 1);
        return JPDSE_OK;
      }
      if (d->stride == 2) {
        // padding resolved by the loader (no padded copy): pays for the stride-2 layers (PatchGAN layer 0)
        t.unpadded = 1;
        t.RH = d->H;
        t.RW = d->W;
        t.r_pad = d->pad;
        t.r_reflect = d->pad_mode == JPDSE_PAD_REFLECT;
        t.AH = p.OH;
        t.AW = p.OW;
      } else {
        // 7x7 stride-1 first convs: 7 filter rows re-read every strip, the per-lane padding arithmetic costs more
        // than one pass of pad_kernel (measured 0.65 vs 0.84 ms)
        if (int rc = launch_pad<T>(x, ws, d->N, d->H, d->W, p.Cs, d->pad, d->pad, d->pad, d->pad, d->pad_mode, s))
          return rc;
        t.XP = reinterpret_cast<const bf16_t*>(ws);
        t.x_limit = (long long)d->N * p.Hp * p.Wp * p.Cs + (long long)(kSlackBytes / 2);
      }
      return launch_wgrad_thin(t, s);
    }
    // the fast kernel's loader uses 32-bit element offsets
    const bool fits32 = (long long)d->N * p.OH * p.OW * p.Ks < (1LL << 31) &&
                        (long long)d->N * p.Hp * p.Wp * p.Cs < (1LL << 31);
    if (g_fast_enabled && p.Ks % 64 == 0 && fits32) {
      FastWgArgs f = {};
      f.DY = reinterpret_cast<const bf16_t*>(dy);
      f.DW = dw;
      f.M = d->N * p.OH * p.OW;
      f.OH = p.OH;
      f.OW = p.OW;
      f.Cs = p.Cs;
      f.C = d->C;
      f.Ks = p.Ks;
      f.K = d->K;
      f.R = d->R;
      f.S = d->S;
      f.sy = f.sx = d->stride;
      if (p.Cs % 64 == 0) {
        f.X = reinterpret_cast<const bf16_t*>(x);
        f.IH = d->H;
        f.IW = d->W;
        f.py = f.px = d->pad;
        f.reflect = d->pad_mode == JPDSE_PAD_REFLECT;
      } else {
        // run mode over the materially padded input (40-channel network inputs, 8-channel images)
        if (!dry)
          if (int rc = launch_pad<T>(x, ws, d->N, d->H, d->W, p.Cs, d->pad, d->pad, d->pad, d->pad, d->pad_mode, s))
            return rc;
        f.X = reinterpret_cast<const bf16_t*>(ws);
        f.IH = p.Hp;
        f.IW = p.Wp;
        f.py = f.px = 0;
        f.reflect = 0;
        f.run_mode = 1;
        f.run_len = d->S * p.Cs;
      }
      return launch_wgrad_fast(f, slabs, s, slab_bytes_out);
    }
  }
  // always staged through the workspace: the GEMM loaders rely on the zeroed slack behind it
  if (!dry)
    if (int rc = launch_pad<T>(x, ws, d->N, d->H, d->W, p.Cs, d->pad, d->pad, d->pad, d->pad, d->pad_mode, s))
      return rc;
  const void* xin = ws;
  GemmWgradArgs a = {};
  a.X = xin;
  a.DY = dy;
  a.DW = dw;
  a.M = d->N * p.OH * p.OW;
  a.OH = p.OH;
  a.OW = p.OW;
  a.K = d->K;
  a.Ks = p.Ks;
  a.C = d->C;
  a.Cs = p.Cs;
  a.R = d->R;
  a.S = d->S;
  a.run = d->S * p.Cs;
  a.in_sn = (long long)p.Hp * p.Wp * p.Cs;
  a.in_sh = (long long)d->stride * p.Wp * p.Cs;
  a.in_sw = (long long)d->stride * p.Cs;
  a.in_sr = (long long)p.Wp * p.Cs;
  a.in_base = 0;
  a.dy_sn = (long long)p.OH * p.OW * p.Ks;
  a.dy_sh = (long long)p.OW * p.Ks;
  a.dy_sw = p.Ks;
  a.dy_base = 0;
  if (dry) {
    *slab_bytes_out = generic_wgrad_slab_bytes<T>(a);
    return JPDSE_OK;
  }
  return launch_wgrad<T>(a, slabs, s);
}

}  // namespace jpdse
