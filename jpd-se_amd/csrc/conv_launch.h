// Launchers of the forward / data-gradient kernel families (generic, fast, halo, tap-program, row-streaming, thin, head),
// the predicates that say which layer takes which kernel, and the in-library GEMM timer.  Every launcher re-checks the
// divisibility / extent assumptions of its kernel's grid before it enqueues anything (a wrong launch argument must be an error
// code, not a GPU memory fault: DESIGN.md 9).  Part of conv_gemm.hip (one translation unit).
#pragma once

namespace jpdse {

// ---- in-library kernel timer (bench.py "roofline"): hipEvent pairs around the GEMM launches whose
// (N, K) signature was selected, recorded on the stream the kernel runs on.
struct GemmProf {
  bool on = false;
  int Ks = 0;
  long long kdim = 0;
  int used = 0;
  std::vector<hipEvent_t> ev;   // 2 per launch
  std::vector<double> flops;
  std::vector<int> cls;         // 0: forward / data-gradient GEMM; 1: reflect ring strips + fold; 2: weight gradient
};
static GemmProf g_prof;
// Launch-argument guard shared by the tiled launchers: the kernels assume an exact tile grid and 32-bit in-tensor offsets;
// the dispatch predicates guarantee both, a launcher that is handed anything else refuses before enqueueing.
static int check_tile_grid(const char* who, int N, int OH, int OW, int th, int tw, int Cs_in, long long in_elems, long long out_elems) {
  if (N <= 0 || OH <= 0 || OW <= 0 || OH % th != 0 || OW % tw != 0)
    return set_error(JPDSE_EINVAL, "%s: output grid %d x %d x %d does not tile into %d x %d patches", who, N, OH, OW, th, tw);
  if (Cs_in <= 0 || Cs_in % 64 != 0) return set_error(JPDSE_EINVAL, "%s: %d input channels (a multiple of 64 is required)", who, Cs_in);
  if (in_elems >= (1LL << 31) || out_elems >= (1LL << 40))
    return set_error(JPDSE_EINVAL, "%s: tensor too large for the kernel's offsets (%lld input elements)", who, in_elems);
  return JPDSE_OK;
}

// regions other than the plain GEMM launches (which record in place): returns the slot or -1
static int prof_begin(hipStream_t s) {
  if (!g_prof.on || (size_t)(2 * g_prof.used + 2) > g_prof.ev.size()) return -1;
  (void)hipEventRecord(g_prof.ev[2 * g_prof.used], s);
  return g_prof.used;
}
static void prof_end(int slot, int cls, double flops, hipStream_t s) {
  if (slot < 0) return;
  (void)hipEventRecord(g_prof.ev[2 * slot + 1], s);
  g_prof.flops[slot] = flops;
  g_prof.cls[slot] = cls;
  g_prof.used = slot + 1;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_fwd_cfg(const GemmFwdArgs& a_in, hipStream_t s) {
  GemmFwdArgs a = a_in;
  const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.Ks + BN - 1) / BN;
  const size_t lds = 2 * (BM + BN) * 64;
  // split-K (fp32, few tiles): only with a slab region big enough behind `partial` (the plan sized it with the same function)
  a.splits = (a.col_mod == 0 && a.partial != nullptr) ? generic_splitk_for((int)sizeof(T), a.M, a.Ks, a.R * a.cpr, a.M / (a.OH * a.OW)) : 1;
  if (a.splits > 1 && (size_t)a.splits * a.M * a.Ks * sizeof(float) > a.partial_cap) a.splits = 1;
  const long long kdim = (long long)a.R * a.cpr * (64 / (int)sizeof(T));
  const bool timed = g_prof.on && a.Ks == g_prof.Ks && kdim == g_prof.kdim &&
                     (size_t)(2 * g_prof.used + 2) <= g_prof.ev.size();
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.used], s);
  hipLaunchKernelGGL((gemm_fwd_kernel<T, BM, BN, WM, WN>), dim3(tiles_m * tiles_n, a.splits > 1 ? a.splits : 1), dim3(64 * WM * WN), lds, s, a);
  int rc = check_launch("gemm_fwd_kernel");
  if (rc == JPDSE_OK && a.splits > 1) {
    const long long total_vec = (long long)a.M * (a.Ks / 4);
    hipLaunchKernelGGL((gemm_splitk_finish_kernel<T>), dim3((unsigned)((total_vec + 255) / 256)), dim3(256), 0, s, a, total_vec);
    rc = check_launch("gemm_splitk_finish_kernel");
  }
  if (timed) {
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s);
    g_prof.flops[g_prof.used] = 2.0 * (double)a.M * (double)a.Ks * (double)kdim;
    g_prof.cls[g_prof.used] = 0;
    ++g_prof.used;
  }
  return rc;
}

template <typename T>
static int launch_fwd(const GemmFwdArgs& a, hipStream_t s) {
  if (a.M <= 0) return JPDSE_OK;
  if (a.Ks > 64) return launch_fwd_cfg<T, 128, 128, 2, 2>(a, s);
  if (a.Ks > 32) return launch_fwd_cfg<T, 128, 64, 2, 2>(a, s);
  return launch_fwd_cfg<T, 256, 32, 4, 1>(a, s);
}

// split count of the generic weight-gradient kernel (shared by the launcher and the workspace query)
template <typename T, int BM, int BN>
static int generic_wgrad_splits(const GemmWgradArgs& a, int* chunks_per_split) {
  const int PIX = WgStage<T>::PIX;
  const int col_tiles = (a.run + BN - 1) / BN, chunks_total = (a.M + PIX - 1) / PIX;
  const int tiles = ((a.K + BM - 1) / BM) * a.R * col_tiles;
  int splits = 1;
  if (tiles < 512) {
    splits = (768 + tiles - 1) / tiles;
    const int max_splits = (chunks_total + 7) / 8;  // >= 8 chunks of work per split
    if (splits > max_splits) splits = max_splits;
    if (splits > 64) splits = 64;
    if (splits < 1) splits = 1;
  }
  const int cps = (chunks_total + splits - 1) / splits;
  if (chunks_per_split) *chunks_per_split = cps;
  return (chunks_total + cps - 1) / cps;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_wgrad_cfg(GemmWgradArgs a, float* slabs, hipStream_t s) {
  const int PIX = WgStage<T>::PIX;
  a.col_tiles_per_r = (a.run + BN - 1) / BN;
  a.chunks_total = (a.M + PIX - 1) / PIX;
  const int tiles = ((a.K + BM - 1) / BM) * a.R * a.col_tiles_per_r;
  const int splits = generic_wgrad_splits<T, BM, BN>(a, &a.chunks_per_split);
  const long long n = (long long)a.K * a.R * a.S * a.C;
  a.partial = splits > 1 ? slabs : nullptr;
  a.slab_stride = (n + 3) / 4 * 4;
  const size_t lds = 2 * (BM + BN) * 64;
  hipLaunchKernelGGL((gemm_wgrad_kernel<T, BM, BN, WM, WN>), dim3(tiles, splits), dim3(64 * WM * WN), lds, s, a);
  if (int rc = check_launch("gemm_wgrad_kernel")) return rc;
  return splits > 1 ? launch_slab_reduce(slabs, a.DW, n, a.slab_stride, splits, s) : JPDSE_OK;
}

template <typename T>
static size_t generic_wgrad_slab_bytes(const GemmWgradArgs& a) {
  const int splits = a.K > 64 ? generic_wgrad_splits<T, 128, 128>(a, nullptr)
                              : (a.K > 32 ? generic_wgrad_splits<T, 64, 128>(a, nullptr) : generic_wgrad_splits<T, 32, 256>(a, nullptr));
  const long long n = (long long)a.K * a.R * a.S * a.C;
  return splits > 1 ? (size_t)splits * ((n + 3) / 4 * 4) * sizeof(float) : 0;
}

template <typename T>
static int launch_wgrad(const GemmWgradArgs& a, float* slabs, hipStream_t s) {
  if (a.K > 64) return launch_wgrad_cfg<T, 128, 128, 2, 2>(a, slabs, s);
  if (a.K > 32) return launch_wgrad_cfg<T, 64, 128, 2, 2>(a, slabs, s);
  return launch_wgrad_cfg<T, 32, 256, 1, 4>(a, slabs, s);
}

JPDSE_SWITCH(int, g_fast_abl, 0);                 // 210: timing-only ablation of the fast kernel (activation tiles staged for one tap in four)
template <int WM, int WN, int TM, int TN, int VAR, int STAGES = 3>
static int launch_fast_cfg(FastBatch& b, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int lds = STAGES * (BM + BN) * 128;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_fast_kernel<WM, WN, TM, TN, VAR, STAGES>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "gemm_fast: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    configured = true;
  }
  int total = 0;
  double flops = 0.0;
  bool timed = g_prof.on && (size_t)(2 * g_prof.used + 2) <= g_prof.ev.size();
  for (int i = 0; i < b.n; ++i) {
    const FastArgs& a = b.p[i];
    if ((a.Y == nullptr && !a.no_finish) || ((a.splits > 1 || a.no_finish) && a.partial == nullptr))
      return set_error(JPDSE_EINVAL, "gemm_fast: problem %d has no output buffer", i);
    if ((a.x_sh ? a.x_sh : (long long)a.IW * a.Cs) * a.IH >= (1LL << 31))
      return set_error(JPDSE_EINVAL, "gemm_fast: one image spans >= 2^31 elements (in-image offsets are 32-bit)");
    if (a.x_extent > 0 && a.OH > 0 && a.OW > 0) {
      // sub-image problems (the ring strips of the reflect data gradient address rows / columns of a larger tensor through
      // x_sn / x_sh): the last element the loader can touch must lie inside the tensor -- a wrong stride or base here is
      // a GPU memory fault, not a wrong number (the round-1 abort)
      const long long n_img = a.M / ((long long)a.OH * a.OW);
      const long long sn = a.x_sn ? a.x_sn : (long long)a.IH * a.IW * a.Cs, sh = a.x_sh ? a.x_sh : (long long)a.IW * a.Cs;
      const long long last = (n_img - 1) * sn + (long long)(a.IH - 1) * sh + (long long)(a.IW - 1) * a.Cs + a.Cs;
      if (n_img < 1 || last > a.x_extent)
        return set_error(JPDSE_EINVAL, "gemm_fast: problem %d addresses element %lld of a %lld-element input", i, last, a.x_extent);
    }
    b.first_tile[i] = total;
    b.p[i].abl = g_fast_abl;
    b.p[i].xcd_map = 0;       // (the XCD-aware tile order, round-3 developer mode 30, measured neutral and was retired: DESIGN.md 4.1 (xi))
    total += ((a.M + BM - 1) / BM) * ((a.Ks + BN - 1) / BN) * (a.splits > 1 ? a.splits : 1);
    const long long kdim = (long long)a.R * a.S * a.Cs;
    flops += 2.0 * (double)a.M * (double)a.Ks * (double)kdim;
    timed = timed && b.n == 1 && a.Ks == g_prof.Ks && kdim == g_prof.kdim;
  }
  for (int i = b.n; i < 5; ++i) b.first_tile[i] = total;
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.used], s);
  hipLaunchKernelGGL((gemm_fast_kernel<WM, WN, TM, TN, VAR, STAGES>), dim3(total), dim3(64 * WM * WN), lds, s, b);
  if (b.n == 1 && b.p[0].splits > 1 && !b.p[0].no_finish) {
    const long long total_vec = (long long)b.p[0].M * (b.p[0].Ks / 8);
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(ew_blocks(total_vec)), dim3(256), 0, s, b.p[0], total_vec);
  }
  if (timed) {
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s);
    g_prof.flops[g_prof.used] = flops;
    g_prof.cls[g_prof.used] = 0;
    ++g_prof.used;
  }
  return check_launch("gemm_fast_kernel");
}

static bool prefer_320(int M, int Ks) {
  // one round of 320-row tiles beats two rounds of 256-row tiles (e.g. the ResnetBlock data gradient
  // on the reflect-padded domain: M = 8976 -> 232 tiles instead of 288 on 256 CUs)
  if (Ks <= 64) return false;
  const long long nt = (Ks + 127) / 128;
  const long long t256 = (long long)((M + 255) / 256) * nt, t320 = (long long)((M + 319) / 320) * nt;
  const long long c256 = ((t256 + 255) / 256) * 256, c320 = ((t320 + 255) / 256) * 320;
  return c320 < c256;
}

// ---- persistent, cross-tile-pipelined form for the short-K layers (gemm_pers.h) -------------------------------------------
// Off since the end of round 4: with the epilogue staging fixed (gemm_fast.h, acc_tile_to_lds) the fast kernel runs the two launches the persistent form
// was shipped for 5-7 % FASTER than it (697 vs 650, 685 vs 640 TFLOP/s; step 24.93 vs 24.95 ms) -- its +8 % had been the lean epilogue it happened to have.
// Developer modes 51 / 52 (tests) / 61 (the round-4 rule: whole-round grids, <= 24 K-tiles) still reach it.
JPDSE_SWITCH(int, g_pers_enabled, 0);
JPDSE_SWITCH(int, g_pers_max_kt, 24);     // K-tile count up to which the persistent form is taken (51: every fast-kernel layer without split-K, A/B)
JPDSE_SWITCH(int, g_pers_min_tiles, 256); // 52: from one tile on (tests: ragged tails and multi-problem launches at small sizes)
static bool pers_ok(const FastBatch& b) {
  if (!g_pers_enabled || b.n <= 0 || b.small_m) return false;
  const int Ks = b.p[0].Ks;
  long long tiles = 0;
  for (int i = 0; i < b.n; ++i) {
    const FastArgs& a = b.p[i];
    const int kt = a.R * a.S * (a.Cs / 64);
    if (a.splits > 1 || a.no_finish || a.mask != nullptr || a.addend != nullptr || a.bias != nullptr || a.act == JPDSE_ACT_TANH) return false;
    if (a.Ks != Ks || a.Ks < 64 || a.Ks % 8 != 0 || a.Cs % 64 != 0 || kt < 4 || kt > g_pers_max_kt || a.M <= 0) return false;
    // 32-bit output offsets in the per-wave epilogue
    const long long n_img = (a.M + (long long)a.OH * a.OW - 1) / ((long long)a.OH * a.OW);
    const long long last = a.out_base + (n_img - 1) * a.out_sn + (long long)(a.OH - 1) * a.out_sh + (long long)(a.OW - 1) * a.out_sw + a.Ks;
    if (a.out_base < 0 || last >= (1LL << 31)) return false;
    tiles += (long long)((a.M + 255) / 256) * ((a.Ks + (Ks > 64 ? 127 : 63)) / (Ks > 64 ? 128 : 64));
  }
  if (g_pers_min_tiles <= 1) return tiles >= 1 && tiles < (1LL << 30);       // developer mode 52 (tests)
  // Measured per layer in one process (profiles/r04_pers_v2_ab.txt): +8-10 % where the 256-row tiles fill whole rounds of the
  // 256 CUs (128 <-> 256 channels, 3x3 stride 2: 1024 tiles), -4 % on PatchGAN layer 1 (1037 tiles: a fifth round for 13 of
  // them), -10 % on layer 2 (526 tiles: 2.05 rounds), neutral at 36 K-tiles.  So: whole rounds only (<= 3 % of idle slots), short loops.
  const long long rounds = (tiles + 255) / 256;
  return tiles >= g_pers_min_tiles && tiles < (1LL << 30) && (rounds * 256 - tiles) * 32 <= tiles;
}

template <int TN>
static int launch_pers_cfg(FastBatch& b, hipStream_t s) {
  constexpr int BN = 2 * TN * 32;
  constexpr int lds = 3 * (256 + BN) * 128;
  static bool configured = false;
  static int cus = 256;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pers_kernel<TN>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "gemm_pers: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) cus = n;
    configured = true;
  }
  int total = 0;
  for (int i = 0; i < b.n; ++i) {
    const FastArgs& a = b.p[i];
    if (a.Y == nullptr) return set_error(JPDSE_EINVAL, "gemm_pers: problem %d has no output buffer", i);
    if ((a.x_sh ? a.x_sh : (long long)a.IW * a.Cs) * a.IH >= (1LL << 31))
      return set_error(JPDSE_EINVAL, "gemm_pers: one image spans >= 2^31 elements (in-image offsets are 32-bit)");
    b.first_tile[i] = total;
    total += ((a.M + 255) / 256) * ((a.Ks + BN - 1) / BN);
  }
  for (int i = b.n; i < 5; ++i) b.first_tile[i] = total;
  const int grid = total < cus ? total : cus;
  hipLaunchKernelGGL((gemm_pers_kernel<TN>), dim3(grid), dim3(512), lds, s, b, total);
  return check_launch("gemm_pers_kernel");
}

static int launch_pers(FastBatch& b, hipStream_t s) {
  return b.p[0].Ks > 64 ? launch_pers_cfg<2>(b, s) : launch_pers_cfg<1>(b, s);
}

JPDSE_SWITCH(int, g_fast_small, 20);      // K-tile count up to which the 128-row / 2-stage fast configs are used
JPDSE_SWITCH(int, g_fast_fill, 1);                // 59: no small tiles for the few-tile medium-K layers (A/B)
static int launch_fast_batch(FastBatch& b, hipStream_t s) {
  if (b.n <= 0) return JPDSE_OK;
  const int Ks = b.p[0].Ks;
  if (b.n == 1 && b.p[0].splits > 1) {   // split-K
    if (Ks > 64) return launch_fast_cfg<4, 2, 2, 2, 0>(b, s);
    if (Ks > 32) return launch_fast_cfg<4, 2, 2, 1, 0>(b, s);
    return launch_fast_cfg<8, 1, 1, 1, 0>(b, s);
  }
  for (int i = 0; i < b.n; ++i)
    if (b.n > 1 && !b.p[i].no_finish) b.p[i].splits = 1;
  if (pers_ok(b)) return launch_pers(b, s);
  if (b.small_m) {
    if (Ks > 64) return launch_fast_cfg<2, 2, 2, 2, 0, 2>(b, s);
    if (Ks > 32) return launch_fast_cfg<2, 2, 2, 1, 0, 2>(b, s);
  }
  int kt = 0;
  for (int i = 0; i < b.n; ++i) {
    const int k = b.p[i].R * b.p[i].S * (b.p[i].Cs / 64);
    kt = k > kt ? k : kt;
  }
  if (g_fast_small && kt <= g_fast_small && b.p[0].splits <= 1 && !g_fast_abl) {
    // short reductions are prologue / epilogue bound: 128-row tiles, 4 waves, 2 stages = 64 (48) KiB of LDS, so two
    // (three) blocks share a CU and overlap each other's fill and store phases
    // (deeper rings for the same tiles -- round-3 developer modes 44 / 45 -- measured 25-50 % slower: profiles/r03_fast_stages_ab.txt)
    // up to 16 K-tiles and wide outputs: 64 x 128 tiles (48 KiB of LDS, three blocks per CU).  Measured per layer in one process
    // (profiles/r03_fast_tile64_ab.txt): PatchGAN layer 1 forward +7 % / +32 % (first / second scale), layer 2 data gradient
    // +10 % / +24 %; longer loops (18-32 tiles) and the 64-wide outputs (64 x 64 tiles) lose 3-6 % and keep the 128-row tiles.
    if (Ks > 64 && kt <= 16) return launch_fast_cfg<2, 2, 1, 2, 0, 2>(b, s);   // 64 x 128
    if (Ks > 64) return launch_fast_cfg<2, 2, 2, 2, 0, 2>(b, s);   // 128 x 128
    if (Ks > 32) return launch_fast_cfg<2, 2, 2, 1, 0, 2>(b, s);   // 128 x 64
  }
  if (g_fast_fill && Ks > 64 && kt <= 64 && b.p[0].splits <= 1 && !g_fast_abl) {
    // Few tiles, medium K (round 4): 256-row tiles of a small layer leave most of the chip idle (PatchGAN layer 2 of the second scale:
    // 17,160 pixels x 256 channels = 136 tiles for 256 CUs, 279 TFLOP/s).  Smaller tiles of the same loop (same summation order) fill it.
    long long t256 = 0;
    for (int i = 0; i < b.n; ++i) t256 += (long long)((b.p[i].M + 255) / 256) * ((Ks + 127) / 128);
    if (t256 < 224) {
      if (t256 < 160) return launch_fast_cfg<2, 2, 1, 2, 0, 2>(b, s);   // 64 x 128
      return launch_fast_cfg<2, 2, 2, 2, 0, 2>(b, s);                    // 128 x 128
    }
  }
  if (b.n == 1 && prefer_320(b.p[0].M, Ks)) return launch_fast_cfg<2, 4, 5, 1, 0, 2>(b, s);   // 320 x 128, 2 stages
#ifdef JPDSE_DEV
  if (Ks > 64 && g_fast_abl) {
    switch (g_fast_abl) {
      case 1: return launch_fast_cfg<4, 2, 2, 2, 1>(b, s);
      case 2: return launch_fast_cfg<4, 2, 2, 2, 2>(b, s);
      case 4: return launch_fast_cfg<4, 2, 2, 2, 4>(b, s);
      case 5: return launch_fast_cfg<4, 2, 2, 2, 5>(b, s);
      case 8: return launch_fast_cfg<4, 2, 2, 2, 8>(b, s);
      case 9: return launch_fast_cfg<4, 2, 2, 2, 9>(b, s);
      case 7: return launch_fast_cfg<4, 2, 2, 2, 7>(b, s);
      case 32: return launch_fast_cfg<4, 2, 2, 2, 32>(b, s);
      case 64: return launch_fast_cfg<4, 2, 2, 2, 64>(b, s);
      case 96: return launch_fast_cfg<4, 2, 2, 2, 96>(b, s);
      case 39: return launch_fast_cfg<4, 2, 2, 2, 39>(b, s);
      case 192: return launch_fast_cfg<4, 2, 2, 2, 192>(b, s);
      case 128: return launch_fast_cfg<4, 2, 2, 2, 128>(b, s);
      default: return launch_fast_cfg<4, 2, 2, 2, 16>(b, s);
    }
  }
#endif
  if (Ks > 64) return launch_fast_cfg<4, 2, 2, 2, 0>(b, s);   // 256 x 128
  if (Ks > 32) return launch_fast_cfg<4, 2, 2, 1, 0>(b, s);   // 256 x 64
  return launch_fast_cfg<8, 1, 1, 1, 0>(b, s);                // 256 x 32
}

static int launch_fast(const FastArgs& a, hipStream_t s) {
  if (a.M <= 0) return JPDSE_OK;
  FastBatch b = {};
  b.p[0] = a;
  b.n = 1;
  return launch_fast_batch(b, s);
}

JPDSE_SWITCH(bool, g_fast_enabled, true);   // jpdse_debug_set_fast_path(0) forces the generic kernels (A/B tests)

// The fast kernel runs ONE 256-row tile per CU (144 KiB of LDS), so its grid should either cover
// the 256 CUs many times over or be an exact multiple of them; in between (e.g. the 288 tiles of the
// ResnetBlock data gradient) the generic 128x128 kernel with 3 co-resident blocks per CU wins
// (measured: scripts/bench_conv.py, profiles/r01_conv_layers_*.log).
static bool prefer_320(int M, int Ks);
static bool fast_pays(int M, int Ks, int k_tiles) {
  if (!g_fast_enabled) return false;
  if (g_pers_min_tiles <= 1 && g_pers_enabled && k_tiles >= 4 && Ks >= 64) return true;   // developer mode 52 (tests): small problems reach the persistent form
  if (k_tiles < 8) return false;   // short reductions (stride-2 sub-pixel phases of 2x2 taps x 64 ch) do not fill the 3-stage ring
  if (Ks <= 32) {
    // measured: the generic 256x32 kernel beats the 8-wave 256x32 fast config on short reductions; with a long
    // one (512 -> 1 PatchGAN map, K = 8192) the fast kernel needs no padded copy and streams the input by DMA
    if (!(g_thin_out_fast && k_tiles >= 64)) return false;
    return splitk_for(M, Ks, k_tiles) > 1 || (M + 255) / 256 >= 128;
  }
  if (splitk_for(M, Ks, k_tiles) > 1) return true;
  const int bn = Ks > 64 ? 128 : (Ks > 32 ? 64 : 32);
  const long long tiles = (long long)((M + 255) / 256) * ((Ks + bn - 1) / bn);
  if (prefer_320(M, Ks)) {
    const long long t320 = (long long)((M + 319) / 320) * ((Ks + 127) / 128);
    if (t320 % 256 == 0 || t320 % 256 >= 192 || t320 >= 448) return true;   // well-filled rounds
  }
  return tiles >= 448 || (tiles >= 256 && tiles % 256 == 0);
}

JPDSE_SWITCH(int, g_ring_enabled, 1);
JPDSE_SWITCH(int, g_ring_virt, 1);        // 40: ring strips as four split-K GEMMs + ring_fold_kernel (the round-1..3 form) instead of the folded frame
JPDSE_SWITCH(int, g_merge_min_kt, 4);
JPDSE_SWITCH(int, g_merge_min_tiles, 64);    // merged stride-phase data gradient on the fast kernel from this many 256-row tiles on (26: 384 as in round 1, A/B)
JPDSE_SWITCH(int, g_halo_single, 1);
JPDSE_SWITCH(int, g_halo_enabled, 1);
JPDSE_SWITCH(int, g_halo_abl, 0);

template <int TN, int ABL = 0, bool SINGLE = false, bool MF16 = false, bool MOM = false, bool VIRT = false, bool NSUM = false>
static int launch_halo_cfg_impl(const HaloArgs& a, hipStream_t s) {
  constexpr int BN = 2 * TN * 32;
  constexpr int UH = ((4 + 2) * (64 + 2) + 7) / 8;
  constexpr int lds = (SINGLE ? 1 : 2) * UH * 1024 + 3 * BN * 128;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_halo_kernel<4, TN, ABL, SINGLE, MF16, MOM, VIRT, NSUM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "gemm_halo: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    configured = true;
  }
  if (int rc = check_tile_grid("gemm_halo", a.N, a.OH, a.OW, 4, 64, a.Cs, (long long)a.N * a.IH * a.IW * a.Cs, (long long)a.N * a.OH * a.OW * a.Ks)) return rc;
  if (VIRT && (a.V == nullptr || a.py != 1 || a.px != 1 || a.IH != a.OH || a.IW != a.OW || a.OH < 8 || a.reflect || (a.Cs < 128) != SINGLE))
    return set_error(JPDSE_EINVAL, "gemm_halo: folded-frame form needs a frame, pad 1, equal grids, >= 8 rows");
  if (MOM && (a.mom == nullptr || a.mom_slots != (a.OH / 4) * (a.OW / 64)))
    return set_error(JPDSE_EINVAL, "gemm_halo: moment epilogue without a moment buffer of %d slots", (a.OH / 4) * (a.OW / 64));
  if (NSUM && (a.nx == nullptr || a.nstats == nullptr || a.nsums == nullptr || a.mom_slots != (a.OH / 4) * (a.OW / 64) ||
               a.out_sw != a.Ks || a.out_sh != (long long)a.OW * a.Ks || a.out_sn != (long long)a.OH * a.OW * a.Ks || a.out_base != 0))
    return set_error(JPDSE_EINVAL, "gemm_halo: norm-backward sums need the norm's input, its statistics, %d slots and a dense output", (a.OH / 4) * (a.OW / 64));
  const int tiles = a.N * (a.OH / 4) * (a.OW / 64) * ((a.Ks + BN - 1) / BN);
  const long long kdim = 9LL * a.Cs;
  const int M = a.N * a.OH * a.OW;
  const bool timed = g_prof.on && a.Ks == g_prof.Ks && kdim == g_prof.kdim &&
                     (size_t)(2 * g_prof.used + 2) <= g_prof.ev.size();
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.used], s);
  hipLaunchKernelGGL((gemm_halo_kernel<4, TN, ABL, SINGLE, MF16, MOM, VIRT, NSUM>), dim3(tiles), dim3(512), lds, s, a);
  if (timed) {
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s);
    g_prof.flops[g_prof.used] = 2.0 * (double)M * (double)a.Ks * (double)kdim;
    g_prof.cls[g_prof.used] = 0;
    ++g_prof.used;
  }
  return check_launch("gemm_halo_kernel");
}

#ifdef JPDSE_DEV
// developer A/B (mode 53): the plain forward launch of the 128-channel-tile halo kernel on the four-wave form (gemm_halo4.h)
static int g_halo4 = 0;
static int launch_halo4(const HaloArgs& a, hipStream_t s) {
  constexpr int UH = ((4 + 2) * (64 + 2) + 7) / 8;
  constexpr int lds = 2 * UH * 1024 + 3 * 128 * 128;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_halo4_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "gemm_halo4: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    configured = true;
  }
  if (int rc = check_tile_grid("gemm_halo4", a.N, a.OH, a.OW, 4, 64, a.Cs, (long long)a.N * a.IH * a.IW * a.Cs, (long long)a.N * a.OH * a.OW * a.Ks)) return rc;
  if (a.Cs < 128 || a.Ks % 128 != 0) return set_error(JPDSE_EINVAL, "gemm_halo4: needs >= 128 input channels and 128-channel output tiles");
  const int tiles = a.N * (a.OH / 4) * (a.OW / 64) * (a.Ks / 128);
  const long long kdim = 9LL * a.Cs;
  const bool timed = g_prof.on && a.Ks == g_prof.Ks && kdim == g_prof.kdim && (size_t)(2 * g_prof.used + 2) <= g_prof.ev.size();
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.used], s);
  hipLaunchKernelGGL(gemm_halo4_kernel, dim3(tiles), dim3(256), lds, s, a);
  if (timed) {
    (void)hipEventRecord(g_prof.ev[2 * g_prof.used + 1], s);
    g_prof.flops[g_prof.used] = 2.0 * (double)a.N * a.OH * a.OW * (double)a.Ks * (double)kdim;
    g_prof.cls[g_prof.used] = 0;
    ++g_prof.used;
  }
  return check_launch("gemm_halo4_kernel");
}
#endif

#ifdef JPDSE_DEV
// developer A/B (mode 56): the same launch on the sixteen-wave form (gemm_halo16.h)
static int launch_halo16(const HaloArgs& a, hipStream_t s) {
  constexpr int UH = ((4 + 2) * (64 + 2) + 7) / 8;
  constexpr int lds = 2 * UH * 1024 + 3 * 128 * 128;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_halo16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "gemm_halo16: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    configured = true;
  }
  if (int rc = check_tile_grid("gemm_halo16", a.N, a.OH, a.OW, 4, 64, a.Cs, (long long)a.N * a.IH * a.IW * a.Cs, (long long)a.N * a.OH * a.OW * a.Ks)) return rc;
  if (a.Cs < 128 || a.Ks % 128 != 0) return set_error(JPDSE_EINVAL, "gemm_halo16: needs >= 128 input channels and 128-channel output tiles");
  const int tiles = a.N * (a.OH / 4) * (a.OW / 64) * (a.Ks / 128);
  hipLaunchKernelGGL(gemm_halo16_kernel, dim3(tiles), dim3(1024), lds, s, a);
  return check_launch("gemm_halo16_kernel");
}
#endif

JPDSE_SWITCH(int, g_halo_xcd, 0);
// One-round grids of the multi-slab form (the ResnetBlock convs: 32 patches x 8 channel tiles on 256 CUs): XCD-aware tile order 2 -- an XCD
// takes 2 channel tiles x half of the patches instead of every 8th tile, so its L2 pulls a quarter of the filter instead of all of it.
// Measured (profiles/r04_halo_xcd_ab.txt): memory-side fetch 172 -> 103 MB per launch, time unchanged (1194-1199 TFLOP/s either way);
// on multi-round and single-slab grids the same order costs 1-4 %, so it is not used there.  60: off (A/B).
JPDSE_SWITCH(int, g_halo_xcd_auto, 1);
JPDSE_SWITCH(int, g_halo_mf16, 0);     // measured: 1020 vs 1032 TFLOP/s on the ResnetBlock conv -- the kernel is not MFMA-clock bound
template <int TN, int ABL = 0>
static int launch_halo_cfg(const HaloArgs& a0, hipStream_t s) {
  HaloArgs a = a0;
  a.xcd_mode = g_halo_xcd;
  {
    const long long blocks = (long long)a.N * (a.OH / 4) * (a.OW / 64) * ((a.Ks + TN * 64 - 1) / (TN * 64));
    if (g_halo_xcd == 0 && g_halo_xcd_auto && a.Cs > 64 && blocks <= 256 && blocks % 8 == 0) a.xcd_mode = 2;
  }
  if (a.V != nullptr) {                   // reflect data gradient with the folded frame
    if constexpr (ABL == 0 && TN == 2) {
      if (a.nsums != nullptr) {           // ... and the sums of the InstanceNorm backward that consumes its output
        if (a.Cs == 64) return set_error(JPDSE_EINVAL, "gemm_halo: norm-backward sums are built for the double-buffered form (>= 128 input channels)");
        return launch_halo_cfg_impl<TN, 0, false, false, false, true, true>(a, s);
      }
    }
    if (a.nsums != nullptr) return set_error(JPDSE_EINVAL, "gemm_halo: norm-backward sums need the 128-channel tile");
    if constexpr (ABL == 0) {
      if (a.Cs == 64) return launch_halo_cfg_impl<TN, 0, true, false, false, true>(a, s);     // one slab: single patch buffer
      return launch_halo_cfg_impl<TN, 0, false, false, false, true>(a, s);
    }
  }
  if (a.mom != nullptr) {                 // conv -> InstanceNorm with the moments in this kernel's epilogue (double-buffered form)
    if constexpr (ABL == 0) return launch_halo_cfg_impl<TN, 0, false, false, true>(a, s);
  }
#ifdef JPDSE_DEV
  if (ABL == 0 && TN == 2 && g_halo4 == 16 && a.Cs >= 128 && a.Ks % 128 == 0 && a.pool == nullptr && a.mask == nullptr && a.addend == nullptr)
    return launch_halo16(a, s);
  if (ABL == 0 && TN == 2 && g_halo4 == 1 && a.Cs >= 128 && a.Ks % 128 == 0 && a.pool == nullptr && a.mask == nullptr && a.addend == nullptr)
    return launch_halo4(a, s);
  if (ABL == 0 && g_halo_mf16) {
    if (a.Cs == 64 && g_halo_single) return launch_halo_cfg_impl<TN, 0, true, true>(a, s);
    return launch_halo_cfg_impl<TN, 0, false, true>(a, s);
  }
#endif
  if (a.Cs == 64 && ABL == 0 && g_halo_single) return launch_halo_cfg_impl<TN, 0, true>(a, s);   // one slab: single patch buffer
  return launch_halo_cfg_impl<TN, ABL, false>(a, s);
}



// band height of the row-streaming kernels: as tall as possible (the filter load and the ring prologue are paid once per block)
// while the grid still fills the chip (`want` blocks); small problems take 16 / 8 / 4
static int rows_band_height(int N, int OH, int strips, int n_tiles, long long want, int min_th = 4) {
  for (int cand = 64; cand >= min_th; cand >>= 1) {
    if (OH % cand != 0) continue;
    if ((long long)N * strips * (OH / cand) * n_tiles >= want) return cand;
  }
  for (int cand = 16; cand >= min_th; cand >>= 1)
    if (OH % cand == 0) return cand;
  return min_th;
}

// 3x3 convs over 64-channel inputs (stride 1 | 2, zero padding): filter in registers, input rows streamed once (conv_rows.h)
JPDSE_SWITCH(int, g_rows_enabled, 1);       // 29: these layers on the halo / fast kernels (A/B)

static bool rows_ok(int R, int S, int stride, int reflect, int act, int OH, int OW, int Cs_in, int Ks_out) {
  return g_fast_enabled && g_rows_enabled && R == 3 && S == 3 && (stride == 1 || stride == 2) && !reflect && Cs_in == 64 &&
         Ks_out % 64 == 0 && OW % 64 == 0 && OH % 4 == 0 &&
         (act == JPDSE_ACT_NONE || act == JPDSE_ACT_RELU || act == JPDSE_ACT_LRELU);
}

template <int STRIDE, int WC, bool FUSED>
static int launch_rows_cfg(RowsArgs a, hipStream_t s) {
  typedef RowsGeom<STRIDE, WC> G;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_rows_kernel<STRIDE, WC, FUSED>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "conv_rows: hipFuncSetAttribute(%d B LDS): %s", G::LDS, hipGetErrorString(e));
    configured = true;
  }
  if (int rc = check_tile_grid("conv_rows", a.N, a.OH, a.OW, 4, 64, 64, (long long)a.N * a.IH * a.IW * 64, (long long)a.N * a.OH * a.OW * a.Ks)) return rc;
  if (a.Ks % (32 * WC) != 0) return set_error(JPDSE_EINVAL, "conv_rows: %d output channels do not split into %d-wide wave tiles", a.Ks, 32 * WC);
  a.n_tiles = a.Ks / (32 * WC);
  a.strips = a.OW / 64;
  a.TH = rows_band_height(a.N, a.OH, a.strips, a.n_tiles, 256LL * ((STRIDE == 1 && WC == 2) ? 2 : 1));
  const int th = a.TH;
  a.bands = a.OH / th;
  a.mom_slots = a.bands * a.strips * (4 / WC);
  const long long blocks = (long long)a.N * a.bands * a.strips * a.n_tiles;
  if (blocks > 0x7fffffffLL) return set_error(JPDSE_EINVAL, "conv_rows: grid too large");
  hipLaunchKernelGGL((conv_rows_kernel<STRIDE, WC, FUSED>), dim3((unsigned)blocks), dim3(256), G::LDS, s, a);
  return check_launch("conv_rows_kernel");
}

static int launch_rows(const RowsArgs& a, int stride, hipStream_t s) {
  const bool fused = a.mask != nullptr || a.addend != nullptr;
  const bool wide = a.Ks % 128 == 0;
  if (stride == 1) {
    if (wide) return fused ? launch_rows_cfg<1, 4, true>(a, s) : launch_rows_cfg<1, 4, false>(a, s);
    return fused ? launch_rows_cfg<1, 2, true>(a, s) : launch_rows_cfg<1, 2, false>(a, s);
  }
  if (wide) return fused ? launch_rows_cfg<2, 4, true>(a, s) : launch_rows_cfg<2, 4, false>(a, s);
  return fused ? launch_rows_cfg<2, 2, true>(a, s) : launch_rows_cfg<2, 2, false>(a, s);
}


// data gradient of the 64 -> 128 3x3 stride-2 conv / forward of the 128 -> 64 ConvTranspose2d at full resolution (dgrad2_rows.h)
JPDSE_SWITCH(int, g_dgrad2_noconf, 0);    // 54: TIMING-ONLY ablation, conflict-free LDS addresses (developer build)
static int launch_dgrad2_rows(Dgrad2Args a, hipStream_t s) {
  typedef Dgrad2Geom G;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad2_rows_kernel<false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
#ifdef JPDSE_DEV
    if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&dgrad2_rows_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
#endif
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "dgrad2_rows: hipFuncSetAttribute(%d B LDS): %s", G::LDS, hipGetErrorString(e));
    configured = true;
  }
  if (int rc = check_tile_grid("dgrad2_rows", a.N, a.OH, a.OW, 4, 64, 128, (long long)a.N * a.OH * a.OW * 128, 4LL * a.N * a.OH * a.OW * 64)) return rc;
  a.strips = a.OW / 64;
  const int th = rows_band_height(a.N, a.OH, a.strips, 1, 256);
  a.TH = th;
  a.bands = a.OH / th;
  a.mom_slots = a.bands * a.strips;
#ifdef JPDSE_DEV
  if (g_dgrad2_noconf) {      // timing-only ablation (developer mode 54): conflict-free LDS addresses, wrong results
    hipLaunchKernelGGL(dgrad2_rows_kernel<true>, dim3((unsigned)(a.N * a.bands * a.strips)), dim3(256), G::LDS, s, a);
    return check_launch("dgrad2_rows_kernel<noconf>");
  }
#endif
  hipLaunchKernelGGL(dgrad2_rows_kernel<false>, dim3((unsigned)(a.N * a.bands * a.strips)), dim3(256), G::LDS, s, a);
  return check_launch("dgrad2_rows_kernel");
}


// 64 -> <= 3 channel heads (7x7 reflect + Tanh; 3x3 zero-pad data gradient of VGG conv1_1) as a row-streaming pass (head_rows.h)
JPDSE_SWITCH(int, g_head_rows32, 1);      // 57: the 32-channel head on head_fwd_kernel, as before round 4 (A/B)
static bool head_rows_ok(const HeadFwdArgs& a, int cin) {
  return g_rows_enabled && (cin == 64 || (cin == 32 && g_head_rows32 && a.R == 7)) && a.K <= 3 && a.Ks_out == 8 && a.OW % 128 == 0 &&
         a.OH % 8 == 0 && a.OH == a.H && a.OW == a.W;
}
template <int R, int CIN = 64>
static int launch_head_rows(const HeadFwdArgs& a, hipStream_t s) {
  typedef HeadRowsGeom<R, CIN> G;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_rows_kernel<R, CIN>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "head_rows: hipFuncSetAttribute(%d B LDS): %s", G::LDS, hipGetErrorString(e));
    configured = true;
  }
  const int strips = a.OW / 128;
  int th = 0;
  for (int cand = 64; cand >= 8; cand >>= 1) {
    if (a.OH % cand != 0) continue;
    if ((long long)a.N * strips * (a.OH / cand) >= 512) { th = cand; break; }
  }
  if (th == 0)
    for (int cand = 16; cand >= 8; cand >>= 1)
      if (a.OH % cand == 0) { th = cand; break; }
  if (th == 0 || strips == 0 || a.OW % 128 != 0)
    return set_error(JPDSE_EINVAL, "head_rows: output grid %d x %d does not tile into bands of >= 8 rows x 128-pixel strips", a.OH, a.OW);
  const int bands = a.OH / th;
  hipLaunchKernelGGL((head_rows_kernel<R, CIN>), dim3((unsigned)(a.N * bands * strips)), dim3(256), G::LDS, s, a, th, bands, strips);
  return check_launch("head_rows_kernel");
}


// data gradient of PatchGAN layer 0 with respect to the image channels (thin_dgrad2_rows.h)
static int launch_thin_dgrad2_rows(ThinDgrad2Args a, hipStream_t s) {
  typedef ThinDgrad2Geom G;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_dgrad2_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "thin_dgrad2_rows: hipFuncSetAttribute(%d B LDS): %s", G::LDS, hipGetErrorString(e));
    configured = true;
  }
  a.strips = a.W / 256;
  int th = 0;
  for (int cand = 64; cand >= 8; cand >>= 1) {
    if (a.H % cand != 0) continue;
    if ((long long)a.N * a.strips * (a.H / cand) >= 512) { th = cand; break; }
  }
  if (th == 0)
    for (int cand = 16; cand >= 8; cand >>= 1)
      if (a.H % cand == 0) { th = cand; break; }
  a.TH = th;
  a.bands = a.H / th;
  hipLaunchKernelGGL(thin_dgrad2_rows_kernel, dim3((unsigned)(a.N * a.bands * a.strips)), dim3(256), G::LDS, s, a);
  return check_launch("thin_dgrad2_rows_kernel");
}


// PatchGAN layer 0 forward (40-channel input, 4x4 stride 2, 64 outputs) as a row-streaming pass (thin_rows.h)
static int launch_thin_rows(ThinFwdArgs a, hipStream_t s) {
  typedef ThinRowsGeom G;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_rows_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "thin_rows: hipFuncSetAttribute(%d B LDS): %s", G::LDS, hipGetErrorString(e));
    configured = true;
  }
  a.tiles_w = (a.OW + 63) / 64;
  // band height: fewest (rounds of 512 blocks: two per CU) x (rows per block + the ~4 rows a block pays for its filter load and prologue)
  int th = 8;
  long long best = -1;
  for (int cand = 64; cand >= 8; cand >>= 1) {
    const long long blocks = (long long)a.N * ((a.OH + cand - 1) / cand) * a.tiles_w;
    const long long cost = ((blocks + 511) / 512) * (cand + 4);
    if (best < 0 || cost < best) { best = cost; th = cand; }
  }
  const int bands = (a.OH + th - 1) / th;
  hipLaunchKernelGGL(thin_rows_kernel, dim3((unsigned)(a.N * bands * a.tiles_w)), dim3(256), G::LDS, s, a, th, bands);
  return check_launch("thin_rows_kernel");
}


// dense 8-channel inputs, 64 outputs, as a row-streaming pass (thin_in_rows.h): the data gradient of the 64 -> 3 7x7 reflect-padded
// head (padded-domain conv with the interior written straight into dx, then the ring fold) and VGG conv1_1 forward (3x3, zero pad)
template <int R, bool DUAL>
static int launch_thin_in_rows(ThinInArgs a, hipStream_t s) {
  typedef ThinInGeom<R> G;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_in_rows_kernel<R, DUAL>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "thin_in_rows: hipFuncSetAttribute(%d B LDS): %s", G::LDS, hipGetErrorString(e));
    configured = true;
  }
  a.strips = (a.OW + 63) / 64;
  int th = 8;
  long long best = -1;
  for (int cand = 64; cand >= 8; cand >>= 1) {
    const long long blocks = (long long)a.N * ((a.OH + cand - 1) / cand) * a.strips;
    const long long cost = ((blocks + 511) / 512) * (cand + 4);
    if (best < 0 || cost < best) { best = cost; th = cand; }
  }
  a.TH = th;
  a.bands = (a.OH + th - 1) / th;
  hipLaunchKernelGGL((thin_in_rows_kernel<R, DUAL>), dim3((unsigned)(a.N * a.bands * a.strips)), dim3(256), G::LDS, s, a);
  if (int rc = check_launch("thin_in_rows_kernel")) return rc;
  if (DUAL) {
    const int band = G::PAD + 1;
    const long long per_img = 2LL * band * a.W + (long long)(a.H - 2 * band) * 2 * band;
    const long long total_vec = (long long)a.N * per_img * (64 / 8);
    hipLaunchKernelGGL((reflect_ring_fold_kernel<bf16_t>), dim3(ew_blocks(total_vec)), dim3(256), 0, s, a.DXP, a.DX, a.N, a.H, a.W,
                       64, G::PAD, total_vec);
    return check_launch("reflect_ring_fold_kernel");
  }
  return JPDSE_OK;
}

// ---- tap-program halo kernel (gemm_taps.h): all four sub-pixel phases of a stride-2 data gradient / ConvTranspose forward
JPDSE_SWITCH(int, g_taps_enabled, 1);       // 35: these layers on the merged-phase fast kernel (A/B)

// Stride-2 data gradients on the tap-program kernel.  3x3 (pad 1, even input): phases (0,0) 2x2 taps, (0,1) 2x1, (1,0) 1x2,
// (1,1) 1x1 over the same dy pixels.  4x4 (pad 2: the PatchGAN layers 1-2, networks.py:430-436): 2x2 taps in every phase; their
// inputs have 2^k + 1 rows / columns, so the phases differ by one row / column: the kernel covers the CORE every phase has
// (a multiple of 4 x 64 sub-pixels), the one-row / one-column FRINGE of the longer phases runs as sub-rectangle problems of the
// fast kernel behind it (fused operands in its epilogue).
JPDSE_SWITCH(int, g_taps_dgrad4_enabled, 1);     // 42: the 4x4 stride-2 data gradients on the merged-phase fast kernel (A/B)
JPDSE_SWITCH(int, g_taps_dgrad4_min_tiles, 1 << 30);   // 43: take the path at any size (developer build: tests, A/B) -- the shipped build never does, see below
struct TapsDgrad2Geom { int core_h, core_w, fringe, taps4; };
static bool taps_dgrad2_geom(const jpdse_conv_desc* d, const ConvPlan& p, const void* mask, const void* addend, const float* mom,
                             TapsDgrad2Geom* g) {
  if (!(g_fast_enabled && g_taps_enabled) || d->dtype != JPDSE_BF16 || d->pad_mode == JPDSE_PAD_REFLECT) return false;
  const bool k3 = d->R == 3 && d->S == 3 && d->pad == 1, k4 = d->R == 4 && d->S == 4 && d->pad == 2 && g_taps_dgrad4_enabled;
  if (d->stride != 2 || !(k3 || k4) || p.nph != 4) return false;
  if (mom != nullptr) return false;
  if (p.Ks % 64 != 0 || p.Ks < 128 || p.Cs % 64 != 0) return false;
  // the kernel's loaders carry 32-bit element offsets into dy and into each phase's panel
  if ((long long)d->N * p.OH * p.OW * p.Ks >= (1LL << 31) || (long long)p.Cs * 4 * p.Ks >= (1LL << 31)) return false;
  int min_h = 1 << 30, min_w = 1 << 30, fringe = 0;
  for (int i = 0; i < 4; ++i) {
    const Phase& f = p.ph[i];
    if (f.Lk != f.Uw * p.Ks) return false;
    if (k3 && (f.Uh != (f.qh == 0 ? 2 : 1) || f.Uw != (f.qw == 0 ? 2 : 1))) return false;
    if (k4 && (f.Uh != 2 || f.Uw != 2)) return false;
    if ((f.Uh - 1) - f.i0h != 0 || (f.Uw - 1) - f.i0w != 0) return false;      // every phase starts at dy pixel (oh, ow)
    min_h = f.cnth < min_h ? f.cnth : min_h;
    min_w = f.cntw < min_w ? f.cntw : min_w;
  }
  const int core_h = min_h / 4 * 4, core_w = min_w / 64 * 64;
  if (core_h < 4 || core_w < 64) return false;
  for (int i = 0; i < 4; ++i) {
    const Phase& f = p.ph[i];
    if (f.cnth > core_h) ++fringe;
    if (f.cntw > core_w) ++fringe;
  }
  if (k3 && fringe != 0) return false;              // the 3x3 layers of the generator have even inputs: no fringe path needed
  // 4x4: measured against the merged-phase fast kernel (profiles/r03_taps_dgrad4_ab.txt).  Bare data gradient, one process:
  // +14 % with 1024 core tiles per program (layer 1 at 257 x 513), +5 % with 256 (layer 2), -9 ... -31 % on the second scale.
  // In the step, where the LeakyReLU mask and the feature-matching addend ride in the epilogue, layer 1 was SLOWER
  // (0.195 + 0.130 ms vs 0.186 + 0.116 ms for the two launches): with one block per CU the operand loads of the two epilogues
  // are exposed, the fast kernel overlaps them across its three co-resident blocks.  These K = 512 ... 1024 loops are
  // prologue / epilogue bound on either kernel; the path stays in the developer build only.
  if (k4 && (long long)d->N * (core_h / 4) * (core_w / 64) * ((p.Cs + 127) / 128) < g_taps_dgrad4_min_tiles) return false;
  if (g != nullptr) *g = {core_h, core_w, fringe, k4 ? 1 : 0};
  (void)mask; (void)addend;
  return true;
}
static bool taps_dgrad2_ok(const jpdse_conv_desc* d, const ConvPlan& p, const void* mask, const void* addend, const float* mom) {
  return taps_dgrad2_geom(d, p, mask, addend, mom, nullptr);
}

template <int TN, int T1A, int T0B, int T1B>
static int launch_taps_dgrad2_cfg(const TapsArgs& a, int total, hipStream_t s) {
  constexpr int PH = 5, PW = 65;
  constexpr int lds = 2 * ((PH * PW + 7) / 8) * 1024 + 3 * (2 * TN * 32) * 128;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_taps_kernel<TN, 4, T1A, T0B, T1B, 1, PH, PW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "gemm_taps: hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    configured = true;
  }
  if (int rc = check_tile_grid("gemm_taps(stride-2 data gradient)", a.N, a.OH, a.OW, 4, 64, a.Cs, (long long)a.N * a.IH * a.IW * a.Cs,
                               4LL * a.N * a.OH * a.OW * a.Ks)) return rc;
  if (total != 2 * a.nblk0 || a.nblk0 != a.N * (a.OH / 4) * (a.OW / 64) * ((a.Ks + 2 * TN * 32 - 1) / (2 * TN * 32)))
    return set_error(JPDSE_EINVAL, "gemm_taps: %d blocks for a grid of 2 x %d", total, a.nblk0);
  hipLaunchKernelGGL((gemm_taps_kernel<TN, 4, T1A, T0B, T1B, 1, PH, PW>), dim3(total), dim3(512), lds, s, a);
  return check_launch("gemm_taps_kernel");
}

static int launch_taps_dgrad2(const jpdse_conv_desc* d, const ConvPlan& p, const void* dy, const void* pack, void* dx, hipStream_t s,
                              const void* mask, const void* addend, float mask_slope = 0.f) {
  TapsDgrad2Geom geo = {};
  if (!taps_dgrad2_geom(d, p, mask, addend, nullptr, &geo)) return set_error(JPDSE_EINVAL, "gemm_taps: not a stride-2 data gradient it covers");
  TapsArgs a = {};
  a.mask = reinterpret_cast<const bf16_t*>(mask);
  a.mask_slope = mask_slope;
  a.addend = reinterpret_cast<const bf16_t*>(addend);
  a.X = reinterpret_cast<const bf16_t*>(dy);
  a.Y = reinterpret_cast<bf16_t*>(dx);
  a.N = d->N;
  a.OH = geo.core_h;
  a.OW = geo.core_w;
  a.IH = p.OH;
  a.IW = p.OW;
  a.Cs = p.Ks;
  a.py = a.px = 0;
  a.Kout = d->C;
  a.Ks = p.Cs;
  a.b_rows = p.Cs;
  a.out_sn = (long long)d->H * d->W * p.Cs;
  a.out_sh = 2LL * d->W * p.Cs;
  a.out_sw = 2LL * p.Cs;
  a.act = JPDSE_ACT_NONE;
  // program 0 = {phase (0,0), phase (1,1)}, program 1 = {phase (0,1), phase (1,0)}: 4 + 1 and 2 + 2 taps (3x3), 4 + 4 twice (4x4)
  const Phase* byq[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};
  for (int i = 0; i < 4; ++i) byq[p.ph[i].qh][p.ph[i].qw] = &p.ph[i];
  const Phase* sets[2][2] = {{byq[0][0], byq[1][1]}, {byq[0][1], byq[1][0]}};
  constexpr int PW = 65;
  for (int g = 0; g < 2; ++g) {
    int t = 0;
    for (int q = 0; q < 2; ++q) {
      const Phase& f = *sets[g][q];
      a.prog[g].B[q] = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + f.pack_off);
      a.prog[g].ktot[q] = (long long)f.Uh * f.Lk;
      a.prog[g].out_base[q] = ((long long)(2 * f.i0h + f.qh - d->pad) * d->W + (2 * f.i0w + f.qw - d->pad)) * p.Cs;
      for (int u = 0; u < f.Uh; ++u)
        for (int w = 0; w < f.Uw; ++w) {
          a.prog[g].tap_off[t] = u * PW + w;
          a.prog[g].tap_koff[t] = u * f.Lk + w * p.Ks;
          ++t;
        }
    }
  }
  const int bn = p.Cs % 128 == 0 ? 128 : 64;
  a.nblk0 = d->N * (a.OH / 4) * (a.OW / 64) * ((p.Cs + bn - 1) / bn);
  int rc;
  if (geo.taps4) rc = bn == 128 ? launch_taps_dgrad2_cfg<2, 4, 4, 4>(a, 2 * a.nblk0, s) : launch_taps_dgrad2_cfg<1, 4, 4, 4>(a, 2 * a.nblk0, s);
  else rc = bn == 128 ? launch_taps_dgrad2_cfg<2, 1, 2, 2>(a, 2 * a.nblk0, s) : launch_taps_dgrad2_cfg<1, 1, 2, 2>(a, 2 * a.nblk0, s);
  if (rc != JPDSE_OK || geo.fringe == 0) return rc;
  // fringe: per phase the sub-pixel rows [core_h, cnth) x all its columns, and the columns [core_w, cntw) x rows [0, core_h)
  FastBatch fb = {};
  auto flush = [&]() -> int {
    if (fb.n == 0) return JPDSE_OK;
    const int r = launch_fast_batch(fb, s);
    fb = FastBatch{};
    return r;
  };
  for (int i = 0; i < 4; ++i) {
    const Phase& f = p.ph[i];
    const int rect[2][4] = {{geo.core_h, 0, f.cnth - geo.core_h, f.cntw}, {0, geo.core_w, geo.core_h, f.cntw - geo.core_w}};   // j0, c0, rows, cols
    for (int q = 0; q < 2; ++q) {
      const int j0 = rect[q][0], c0 = rect[q][1], rows = rect[q][2], cols = rect[q][3];
      if (rows <= 0 || cols <= 0) continue;
      FastArgs g = {};
      g.X = a.X;
      g.B = reinterpret_cast<const bf16_t*>(reinterpret_cast<const char*>(pack) + f.pack_off);
      g.Y = a.Y;
      g.M = d->N * rows * cols;
      g.OH = rows;
      g.OW = cols;
      g.IH = p.OH;
      g.IW = p.OW;
      g.Cs = p.Ks;
      g.R = f.Uh;
      g.S = f.Uw;
      g.sy = g.sx = 1;
      g.py = (f.Uh - 1) - f.i0h - j0;
      g.px = (f.Uw - 1) - f.i0w - c0;
      g.Kout = d->C;
      g.Ks = p.Cs;
      g.b_rows = p.Cs;
      g.out_sn = a.out_sn;
      g.out_sh = a.out_sh;
      g.out_sw = a.out_sw;
      g.out_base = ((long long)(2 * (f.i0h + j0) + f.qh - d->pad) * d->W + (2 * (f.i0w + c0) + f.qw - d->pad)) * p.Cs;
      g.act = JPDSE_ACT_NONE;
      g.mask = a.mask;
      g.mask_slope = mask_slope;
      g.addend = a.addend;
      g.splits = 1;
      fb.p[fb.n++] = g;
      if (fb.n == 4)
        if (int r = flush()) return r;
    }
  }
  return flush();
}

// ---- 4x4 stride-1 zero-padded convs and their (single-phase) data gradient on the tap-program kernel: PatchGAN layer 3 of
// both scales (networks.py:430-449).  Their grids are odd (66 x 130, 34 x 66): the kernel's 8 x 32 tiles cover the CORE
// (64 x 128: 95 % of the pixels) with the 11 x 35 input patch staged once per 64-channel slab for all 16 taps; the fringe (the
// last OH % 8 rows, the last OW % 32 columns) runs as two sub-rectangle problems of ONE split-K launch of gemm_fast_kernel
// (fp32 slabs, fixed summation order) + its finish kernels.
JPDSE_SWITCH(int, g_taps4_enabled, 1);      // 36: these layers on the fast kernel alone (A/B)

struct Taps4View {            // a stride-1 4x4 conv as the kernels see it: forward, or the data gradient over dy
  const bf16_t* X; const bf16_t* B; const float* bias; bf16_t* Y;
  int N, IH, IW, Cin_s, OH, OW, py, px, Kout, Ks_out;
  long long ktot;             // panel row stride (elements)
  int tap_r, tap_s;           // panel offsets per filter-row / filter-column step
  int act; float slope;
  const bf16_t* addend; const bf16_t* mask;   // optional fused operands of a data gradient (Y's addressing)
  const bf16_t* frame;                        // nine-tap program only: folded frame of X (reflect data gradient in one launch)
  int reflect;                                // 3x3 view only: mirrored instead of zero padding
};

static bool taps4_shape_ok(int R, int S, int stride, int OH, int OW, int Cin_s, int Ks_out, long long x_elems, long long b_elems) {
  return g_fast_enabled && g_taps4_enabled && R == 4 && S == 4 && stride == 1 && OH >= 8 && OW >= 32 && Cin_s % 64 == 0 &&
         Cin_s >= 128 && Ks_out % 64 == 0 && Ks_out >= 64 && x_elems < (1LL << 31) && b_elems < (1LL << 31);
}

static int taps4_fringe_splits(int N, int OH, int OW, int Ks_out, int k_tiles) {
  const int OHc = OH / 8 * 8, OWc = OW / 32 * 32;
  const long long m_bot = (long long)N * (OH - OHc) * OW, m_right = (long long)N * OHc * (OW - OWc);
  const long long nt = (Ks_out + 127) / 128;
  const long long tiles = ((m_bot + 255) / 256 + (m_right + 255) / 256) * nt;
  if (tiles <= 0) return 0;
  long long sp = 256 / tiles;
  if (sp > k_tiles / 8) sp = k_tiles / 8;
  if (sp > 8) sp = 8;
  return sp < 1 ? 1 : (int)sp;
}

static size_t taps4_fringe_bytes(int N, int OH, int OW, int Ks_out, int k_tiles) {
  const int OHc = OH / 8 * 8, OWc = OW / 32 * 32;
  const long long m = (long long)N * (OH - OHc) * OW + (long long)N * OHc * (OW - OWc);
  return (size_t)taps4_fringe_splits(N, OH, OW, Ks_out, k_tiles) * m * Ks_out * sizeof(float);
}

template <int TN>
static int launch_taps4_cfg(const TapsArgs& a, int total, hipStream_t s) {
  constexpr int PH = 11, PW = 35;
  constexpr int lds = 2 * ((PH * PW + 7) / 8) * 1024 + 3 * (2 * TN * 32) * 128;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_taps_kernel<TN, 16, 0, 0, 0, 2, PH, PW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "gemm_taps(4x4): hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    configured = true;
  }
  if (int rc = check_tile_grid("gemm_taps(4x4)", a.N, a.OH, a.OW, 8, 32, a.Cs, (long long)a.N * a.IH * a.IW * a.Cs,
                               (long long)a.N * a.OH * a.OW * a.Ks)) return rc;
  if (total != a.N * (a.OH / 8) * (a.OW / 32) * ((a.Ks + 2 * TN * 32 - 1) / (2 * TN * 32)))
    return set_error(JPDSE_EINVAL, "gemm_taps(4x4): %d blocks do not match the tile grid", total);
  hipLaunchKernelGGL((gemm_taps_kernel<TN, 16, 0, 0, 0, 2, PH, PW>), dim3(total), dim3(512), lds, s, a);
  return check_launch("gemm_taps_kernel(4x4)");
}

static int launch_taps4(const Taps4View& v, void* ws, hipStream_t s) {
  const int OHc = v.OH / 8 * 8, OWc = v.OW / 32 * 32;
  TapsArgs a = {};
  a.X = v.X;
  a.Y = v.Y;
  a.bias = v.bias;
  a.N = v.N;
  a.OH = OHc;
  a.OW = OWc;
  a.IH = v.IH;
  a.IW = v.IW;
  a.Cs = v.Cin_s;
  a.py = v.py;
  a.px = v.px;
  a.Kout = v.Kout;
  a.Ks = v.Ks_out;
  a.b_rows = v.Ks_out;
  a.out_sn = (long long)v.OH * v.OW * v.Ks_out;
  a.out_sh = (long long)v.OW * v.Ks_out;
  a.out_sw = v.Ks_out;
  a.act = v.act;
  a.slope = v.slope;
  a.addend = v.addend;
  a.mask = v.mask;
  a.prog[0].B[0] = v.B;
  a.prog[0].ktot[0] = v.ktot;
  a.prog[0].out_base[0] = 0;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      a.prog[0].tap_off[r * 4 + c] = r * 35 + c;
      a.prog[0].tap_koff[r * 4 + c] = r * v.tap_r + c * v.tap_s;
    }
  const int bn = v.Ks_out % 128 == 0 ? 128 : 64;
  a.nblk0 = v.N * (OHc / 8) * (OWc / 32) * ((v.Ks_out + bn - 1) / bn);
  if (int rc = bn == 128 ? launch_taps4_cfg<2>(a, a.nblk0, s) : launch_taps4_cfg<1>(a, a.nblk0, s)) return rc;
  // fringe: bottom rows [OHc, OH) x all columns, right columns [OWc, OW) x rows [0, OHc)
  const int k_tiles = 16 * v.Cin_s / 64;
  const int sp = taps4_fringe_splits(v.N, v.OH, v.OW, v.Ks_out, k_tiles);
  if (sp == 0) return JPDSE_OK;
  FastBatch fb = {};
  float* slab = reinterpret_cast<float*>(ws);
  const int rect[2][4] = {{OHc, 0, v.OH - OHc, v.OW}, {0, OWc, OHc, v.OW - OWc}};     // oh0, ow0, rows, cols
  for (int q = 0; q < 2; ++q) {
    const int oh0 = rect[q][0], ow0 = rect[q][1], rows = rect[q][2], cols = rect[q][3];
    if (rows <= 0 || cols <= 0) continue;
    FastArgs g = {};
    g.X = v.X;
    g.B = v.B;
    g.bias = v.bias;
    g.Y = v.Y;
    g.M = v.N * rows * cols;
    g.OH = rows;
    g.OW = cols;
    g.IH = v.IH;
    g.IW = v.IW;
    g.Cs = v.Cin_s;
    g.R = g.S = 4;
    g.sy = g.sx = 1;
    g.py = v.py - oh0;
    g.px = v.px - ow0;
    g.reflect = 0;
    g.Kout = v.Kout;
    g.Ks = v.Ks_out;
    g.b_rows = v.Ks_out;
    g.out_sn = a.out_sn;
    g.out_sh = a.out_sh;
    g.out_sw = a.out_sw;
    g.out_base = ((long long)oh0 * v.OW + ow0) * v.Ks_out;
    g.act = v.act;
    g.slope = v.slope;
    g.splits = sp;
    g.no_finish = 1;
    g.partial = slab;
    g.b_stride = v.ktot;
    g.b_tap_r = v.tap_r;
    g.b_tap_s = v.tap_s;
    g.addend = v.addend;        // applied by splitk_finish_kernel
    g.mask = v.mask;
    slab += (size_t)sp * g.M * v.Ks_out;
    fb.p[fb.n++] = g;
  }
  if (int rc = launch_fast_batch(fb, s)) return rc;
  for (int q = 0; q < fb.n; ++q) {
    const long long total_vec = (long long)fb.p[q].M * (fb.p[q].Ks / 8);
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(ew_blocks(total_vec)), dim3(256), 0, s, fb.p[q], total_vec);
  }
  return check_launch("taps4 fringe finish");
}

// ---- 3x3 stride-1 convs whose grid does not tile into the halo kernel's 4 x 64 patches but into 8 x 32 ones (W = 32: the
// 1024-channel ResnetBlocks of the LocalEnhancer trunk at 16 x 32 pixels, BASELINE config 3): the tap-program kernel with nine
// taps, zero or mirrored padding in the patch loader, and -- few tiles, K = 9216 -- split-K over the channel slabs.
JPDSE_SWITCH(int, g_taps9_enabled, 1);      // 38: these layers on the split-K fast kernel (A/B)

static bool taps9_shape_ok(int R, int S, int stride, int OH, int OW, int Cin_s, int Ks_out, long long x_elems, long long b_elems) {
  return g_fast_enabled && g_taps9_enabled && R == 3 && S == 3 && stride == 1 && OH % 8 == 0 && OW % 32 == 0 && OW % 64 != 0 &&
         Cin_s % 64 == 0 && Cin_s >= 128 && Ks_out % 64 == 0 && x_elems < (1LL << 31) && b_elems < (1LL << 31);
}

static int taps9_splits(int N, int OH, int OW, int Cin_s, int Ks_out) {
  const int sp = splitk_for(N * OH * OW, Ks_out, 9 * Cin_s / 64);      // the split-K fast path's choice: its workspace region is reused
  const int cc = Cin_s / 64;
  return sp > cc / 2 ? (cc / 2 > 0 ? cc / 2 : 1) : sp;                 // >= 2 slabs (18 tap steps) per block
}

template <int TN, bool VIRT = false>
static int launch_taps9_cfg(const TapsArgs& a, int total, hipStream_t s) {
  constexpr int PH = 10, PW = 34;
  constexpr int lds = 2 * ((PH * PW + 7) / 8) * 1024 + 3 * (2 * TN * 32) * 128;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_taps_kernel<TN, 9, 0, 0, 0, 2, PH, PW, VIRT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "gemm_taps(3x3): hipFuncSetAttribute(%d B LDS): %s", lds, hipGetErrorString(e));
    configured = true;
  }
  if (int rc = check_tile_grid("gemm_taps(3x3)", a.N, a.OH, a.OW, 8, 32, a.Cs, (long long)a.N * a.IH * a.IW * a.Cs,
                               (long long)a.N * a.OH * a.OW * a.Ks)) return rc;
  const int sp = a.splits > 1 ? a.splits : 1;
  if (total != sp * a.N * (a.OH / 8) * (a.OW / 32) * ((a.Ks + 2 * TN * 32 - 1) / (2 * TN * 32)) || (sp > 1 && a.partial == nullptr))
    return set_error(JPDSE_EINVAL, "gemm_taps(3x3): %d blocks do not match the tile grid x %d splits", total, sp);
  if (VIRT && (a.V == nullptr || a.py != 1 || a.px != 1 || a.IH != a.OH || a.IW != a.OW || a.OH < 16 || a.reflect))
    return set_error(JPDSE_EINVAL, "gemm_taps(3x3): folded-frame form needs a frame, pad 1, equal grids, >= 16 rows");
  hipLaunchKernelGGL((gemm_taps_kernel<TN, 9, 0, 0, 0, 2, PH, PW, VIRT>), dim3(total), dim3(512), lds, s, a);
  return check_launch("gemm_taps_kernel(3x3)");
}

// `slabs`: fp32 workspace of >= splits * N OH OW * Ks_out floats (the plan's split-K region)
static int launch_taps9(const Taps4View& v, float* slabs, hipStream_t s) {
  TapsArgs a = {};
  a.X = v.X;
  a.Y = v.Y;
  a.bias = v.bias;
  a.N = v.N;
  a.OH = v.OH;
  a.OW = v.OW;
  a.IH = v.IH;
  a.IW = v.IW;
  a.Cs = v.Cin_s;
  a.py = v.py;
  a.px = v.px;
  a.reflect = v.reflect;
  a.Kout = v.Kout;
  a.Ks = v.Ks_out;
  a.b_rows = v.Ks_out;
  a.out_sn = (long long)v.OH * v.OW * v.Ks_out;
  a.out_sh = (long long)v.OW * v.Ks_out;
  a.out_sw = v.Ks_out;
  a.act = v.act;
  a.slope = v.slope;
  a.addend = v.addend;
  a.mask = v.mask;
  a.V = v.frame;
  a.prog[0].B[0] = v.B;
  a.prog[0].ktot[0] = v.ktot;
  a.prog[0].out_base[0] = 0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) {
      a.prog[0].tap_off[r * 3 + c] = r * 34 + c;
      a.prog[0].tap_koff[r * 3 + c] = r * v.tap_r + c * v.tap_s;
    }
  const int bn = v.Ks_out % 128 == 0 ? 128 : 64;
  const int tiles = v.N * (v.OH / 8) * (v.OW / 32) * ((v.Ks_out + bn - 1) / bn);
  a.splits = taps9_splits(v.N, v.OH, v.OW, v.Cin_s, v.Ks_out);
  a.partial = a.splits > 1 ? slabs : nullptr;
  a.nblk0 = tiles * (a.splits > 1 ? a.splits : 1);
  int rc;
  if (v.frame != nullptr) rc = bn == 128 ? launch_taps9_cfg<2, true>(a, a.nblk0, s) : launch_taps9_cfg<1, true>(a, a.nblk0, s);
  else rc = bn == 128 ? launch_taps9_cfg<2>(a, a.nblk0, s) : launch_taps9_cfg<1>(a, a.nblk0, s);
  if (rc) return rc;
  if (a.splits <= 1) return JPDSE_OK;
  FastArgs f = {};                                  // what splitk_finish_kernel reads
  f.Y = v.Y;
  f.bias = v.bias;
  f.M = v.N * v.OH * v.OW;
  f.OH = v.OH;
  f.OW = v.OW;
  f.Kout = v.Kout;
  f.Ks = v.Ks_out;
  f.out_sn = a.out_sn;
  f.out_sh = a.out_sh;
  f.out_sw = a.out_sw;
  f.out_base = 0;
  f.act = v.act;
  f.slope = v.slope;
  f.splits = a.splits;
  f.partial = slabs;
  f.addend = v.addend;
  f.mask = v.mask;
  const long long total_vec = (long long)f.M * (f.Ks / 8);
  hipLaunchKernelGGL(splitk_finish_kernel, dim3(ew_blocks(total_vec)), dim3(256), 0, s, f, total_vec);
  return check_launch("taps9 split-K finish");
}

// 3x3 stride-1 convs whose output grid tiles into 4 x 64 patches (ResnetBlocks, VGG19, and the data
// gradient of the zero-padded ones): LDS-resident input halo, see gemm_halo.h
static bool halo_ok(int R, int S, int stride, int OH, int OW, int Cs_in, int Ks_out) {
  return g_fast_enabled && g_halo_enabled && R == 3 && S == 3 && stride == 1 && OH % 4 == 0 && OW % 64 == 0 &&
         Cs_in % 64 == 0 && Ks_out > 32;
}

// Convs with K*R*S <= 32 outputs-times-taps (the 512 -> 1 PatchGAN map): y[p][k] = sum_taps Z[p + tap][k, tap] with
// Z[q][(k, tap)] = sum_c x[q][c] * w[k][tap][c] -- a 1x1 GEMM over the INPUT pixels (K*R*S <= 32 columns: one MFMA
// tile, every input pixel read once, no padded copy) followed by this gather-sum over the taps.  The direct
// form wastes 31 of 32 MFMA columns and re-reads the input once per tap.
__global__ __launch_bounds__(256) void tapsum_kernel(const float* __restrict__ Z, const float* __restrict__ bias,
                                                    bf16_t* __restrict__ y, int N, int H, int W, int OH, int OW,
                                                    int K, int Ks_out, int R, int S, int pad, int reflect, int zs,
                                                    int act, float slope, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over output pixels x Ks_out
  if (idx >= total) return;
  const int k = (int)(idx % Ks_out);
  long long t = idx / Ks_out;
  const int ow = (int)(t % OW);
  t /= OW;
  const int oh = (int)(t % OH), n = (int)(t / OH);
  float v = 0.f;
  if (k < K) {
    v = bias != nullptr ? bias[k] : 0.f;
    for (int r = 0; r < R; ++r) {
      int ih = oh + r - pad;
      if (reflect) ih = ih < 0 ? -ih : (ih >= H ? 2 * (H - 1) - ih : ih);
      else if ((unsigned)ih >= (unsigned)H) continue;
      for (int s2 = 0; s2 < S; ++s2) {
        int iw = ow + s2 - pad;
        if (reflect) iw = iw < 0 ? -iw : (iw >= W ? 2 * (W - 1) - iw : iw);
        else if ((unsigned)iw >= (unsigned)W) continue;
        v += Z[(((long long)n * H + ih) * W + iw) * zs + (k * R + r) * S + s2];
      }
    }
    v = apply_act(v, act, slope);
  }
  y[idx] = f2bf(v);
}

JPDSE_SWITCH(int, g_thin_fwd_enabled, 1);
// geometry of the thin forward kernel for a layer: TH output rows per block (8, or 4 for stride 2 / when LDS is short)
struct ThinFwdGeom { int TH, TW, strip_units, w_units, lds; };
static bool thin_fwd_geom(const jpdse_conv_desc* d, const ConvPlan& p, ThinFwdGeom* g) {
  if (!(g_fast_enabled && g_thin_fwd_enabled && p.thinf)) return false;
  const int st = d->stride;
  g->w_units = (p.Ks * p.KP_thin * 2 + 1023) / 1024;
  static const int cand[3][2] = {{8, 64}, {4, 64}, {4, 32}};
  // 8-channel inputs (VGG conv1_1) are output-write bound: the smaller block keeps the epilogue tile at 48 KiB so that
  // three blocks share a CU
  for (int c = (p.Cs <= 8 && p.Ks == 64) ? 1 : 0; c < 3; ++c) {
    const int TH = cand[c][0], TW = cand[c][1];
    if (TH == 4 && p.Ks != 64) break;                 // 4 rows x 2 column groups needs K = 64 (32 per group)
    g->strip_units = (((TW - 1) * st + d->S) * p.Cs * 2 + 16 + 1023) / 1024;
    const int lds = (((TH - 1) * st + d->R) * g->strip_units + 2 * g->w_units) * 1024;
    const int epi = TH * TW * (p.Ks * 2 + 64);
    g->TH = TH;
    g->TW = TW;
    g->lds = lds > epi ? lds : epi;
    if (g->lds <= 160 * 1024) return true;
  }
  return false;
}

template <int TN, int TH, int ST, int TW>
static int launch_thin_fwd(const ThinFwdArgs& a, int lds, hipStream_t s) {
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&thin_fwd_kernel<TN, TH, ST, TW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "thin_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((thin_fwd_kernel<TN, TH, ST, TW>), dim3(a.N * a.tiles_h * a.tiles_w), dim3(512), lds, s, a);
  return check_launch("thin_fwd_kernel");
}

JPDSE_SWITCH(int, g_head_fwd_enabled, 1);
static bool head_fwd_ok(const jpdse_conv_desc* d, const ConvPlan& p) {
  const int ncols = d->K * d->R * d->S;
  return g_fast_enabled && g_head_fwd_enabled && p.ES == 2 && d->stride == 1 && d->K <= 3 && p.Ks == 8 &&
         (p.Cs == 64 || p.Cs == 32) && d->R == 7 && d->S == 7 && ncols <= 160 && p.Lk_fwd == d->S * p.Cs;
}

template <int CIN, int NT = 5, int FR = 7, int FS = 7>
static int launch_head_fwd(const HeadFwdArgs& a, hipStream_t s) {
  constexpr int lds = NT * 32 * CIN * 2 + 3 * kHeadMR * CIN * 2 + NT * 32 * kHeadZP * 4 + kHeadTH * 64 * 4 * 4;
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&head_fwd_kernel<CIN, NT, FR, FS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return set_error(JPDSE_ELAUNCH, "head_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((head_fwd_kernel<CIN, NT, FR, FS>), dim3(a.N * a.tiles_h * a.tiles_w), dim3(64 * NT), lds, s, a);
  return check_launch("head_fwd_kernel");
}

JPDSE_SWITCH(int, g_tapsum_enabled, 1);
static bool tapsum_ok(const jpdse_conv_desc* d, const ConvPlan& p) {
  return g_fast_enabled && g_tapsum_enabled && p.ES == 2 && d->stride == 1 && d->K * d->R * d->S <= 32 &&
         p.Cs % 64 == 0 && p.Cs >= 256 && p.Lk_fwd == d->S * p.Cs;
}


// ---- forward moments for the InstanceNorm that follows a conv (jpdse_conv_fwd_moments): which layers write them, and how many
// slots per image.  These mirror the dispatch order of conv_fwd_t / conv_dgrad_t.
static bool thin_rows_takes(const jpdse_conv_desc* d, const ConvPlan& p) {
  return g_rows_enabled && d->stride == 2 && d->R == 4 && d->S == 4 && p.Cs == 40 && p.Ks == 64 && d->K == 64 &&
         d->pad_mode != JPDSE_PAD_REFLECT && p.KP_thin == 168;
}
static bool thin_in_rows_takes(const jpdse_conv_desc* d, const ConvPlan& p) {
  return g_fast_enabled && g_rows_enabled && p.Cs == 8 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 &&
         d->pad_mode == JPDSE_PAD_ZERO && p.Ks == 64 && d->K == 64 && p.Lk_fwd == 32;
}
JPDSE_SWITCH(int, g_moments_fused, 1);     // 32 (and 6: the rounding-point-preserving comparison mode): no moment epilogues
static int conv_fwd_moment_slots(const jpdse_conv_desc* d, const ConvPlan& p) {
  if (!g_moments_fused || d->dtype != JPDSE_BF16 || d->act != JPDSE_ACT_NONE) return 0;
  if (thin_in_rows_takes(d, p)) return 0;
  ThinFwdGeom tg;
  if (thin_fwd_geom(d, p, &tg)) {
    if (thin_rows_takes(d, p)) return 0;
    return (p.OH % tg.TH == 0 && p.OW % tg.TW == 0) ? (p.OH / tg.TH) * (p.OW / tg.TW) : 0;
  }
  if (head_fwd_ok(d, p) || tapsum_ok(d, p)) return 0;
  if (rows_ok(d->R, d->S, d->stride, d->pad_mode == JPDSE_PAD_REFLECT, d->act, p.OH, p.OW, p.Cs, p.Ks)) {
    const int WC = p.Ks % 128 == 0 ? 4 : 2, strips = p.OW / 64, n_tiles = p.Ks / (32 * WC);
    const int th = rows_band_height(d->N, p.OH, strips, n_tiles, 256LL * ((d->stride == 1 && WC == 2) ? 2 : 1));
    return (p.OH / th) * strips * (4 / WC);
  }
  if (d->pad_mode != JPDSE_PAD_REFLECT && p.Lk_fwd == d->S * p.Cs &&
      taps4_shape_ok(d->R, d->S, d->stride, p.OH, p.OW, p.Cs, p.Ks, (long long)d->N * d->H * d->W * p.Cs, (long long)p.Ks * 16 * p.Cs))
    return 0;
  if (p.Lk_fwd == d->S * p.Cs &&
      taps9_shape_ok(d->R, d->S, d->stride, p.OH, p.OW, p.Cs, p.Ks, (long long)d->N * d->H * d->W * p.Cs, (long long)p.Ks * 9 * p.Cs))
    return 0;
  // halo kernel (double-buffered form: inputs of 128+ channels): one slot per 4 x 64 output patch
  if (halo_ok(d->R, d->S, d->stride, p.OH, p.OW, p.Cs, p.Ks) && p.Cs > 64 && g_halo_abl == 0) return (p.OH / 4) * (p.OW / 64);
  return 0;
}
static bool dgrad2_rows_takes(const jpdse_conv_desc* d, const ConvPlan& p) {
  return d->dtype == JPDSE_BF16 && g_fast_enabled && g_rows_enabled && d->pad_mode != JPDSE_PAD_REFLECT && d->stride == 2 && d->R == 3 &&
         d->S == 3 && d->pad == 1 && p.Ks == 128 && p.Cs == 64 && d->C == 64 && d->H == 2 * p.OH && d->W == 2 * p.OW &&
         p.OW % 64 == 0 && p.OH % 4 == 0 && p.nph == 4;
}
static int convT_fwd_moment_slots(const jpdse_conv_desc* d, const ConvPlan& p) {
  if (!g_moments_fused || !dgrad2_rows_takes(d, p)) return 0;
  const int strips = p.OW / 64;
  return (p.OH / rows_band_height(d->N, p.OH, strips, 1, 256)) * strips;
}

}  // namespace jpdse
