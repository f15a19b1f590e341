// 3x3 stride-1 convolution (forward, and the data gradient of zero-padded 3x3 convs) with an
// LDS-resident input halo -- the kernel for the ResnetBlock convs and VGG19.
//
// Why: hardware counters on gemm_fast_kernel show 0 LDS bank conflicts, the MFMA pipe ~42 % busy and
// waves parked ~43 % of the time at s_waitcnt/s_barrier: a 256x128x64 K-tile needs 48 KiB of L2->LDS
// fill per 1024 MFMA cycles, above what one CU's LDS-DMA path sustains (~70 GB/s, MI355X_MICROARCH.md
// "Indexed rows: gather into LDS").  2/3 of that fill is the activation tile, and for a 3x3 filter
// the nine taps read the SAME pixels shifted by one.  Here a block owns a TH x 64 output patch of one
// image: per 64-channel slab it stages the (TH+2) x 66 input patch ONCE (rows of 128 B = full lines) (padding resolved while
// loading), then loops over the 9 taps streaming only the 16 KiB weight tile; the A fragments are
// read from the patch at tap-shifted addresses.  Fill traffic per tap: 21.6 KiB instead of 48 KiB.
//
// Pipeline: weight tiles in a 3-stage ring (counted vmcnt, one raw barrier per tap); the NEXT slab's
// patch is double buffered and its DMA units are spread over taps 0..6 of the current slab.
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include <type_traits>

namespace jpdse {

// ds_read_b128 as inline asm (invisible to hipcc's wait insertion: the caller counts lgkmcnt by hand)
__device__ __forceinline__ s16x8 lds_read128_asm(uint32_t addr) {
  s16x8 v;
  asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
  return v;
}
struct HaloArgs {
  const bf16_t* X;   // [N][IH][IW][Cs]
  const bf16_t* B;   // panel [b_rows][9*Cs]
  const float* bias;
  bf16_t* Y;
  int N, OH, OW;     // output grid; requires OH % TH == 0, OW % 64 == 0
  int IH, IW, Cs;
  int py, px, reflect;
  int Kout, Ks, b_rows;
  long long out_sn, out_sh, out_sw, out_base;
  int act;
  float slope;
  const bf16_t* mask;   // optional fused ReLU backward (see FastArgs::mask)
  const bf16_t* addend; // optional: Y = result + addend
  int xcd_mode;         // 0: block b -> tile b; 1: blocks of one XCD (b % 8) take consecutive tiles; 2: 2 N-tiles x half the patches per XCD; 3: 4 N-tiles x a quarter of the patches (one-round grids of 32 x 8 tiles)
  bf16_t* pool;         // optional: MaxPool2d(2, 2) of Y, [N][OH/2][OW/2][Ks], written from the same epilogue tile (VGG19 conv -> ReLU -> pool)
  const bf16_t* V;      // VIRT instantiation: the folded frame of the input (see ring_frame_kernel), [N][2(IW+2) + 2 IH][Cs]
  float* mom;           // optional (MOM instantiation): InstanceNorm moments of y, one (mean, M2) slot per block and channel
  int mom_slots;        //   [N][Ks][mom_slots][2], slot = the block's patch index inside its image (common.h; no bias / activation)
  // NSUM instantiation (round 4): Y is the gradient dy w.r.t. the OUTPUT of an InstanceNorm (+ activation) whose input was nx
  // (same addressing as Y) with statistics nstats [N][Ks][2] = (mean, rstd).  The epilogue also writes, per block and channel,
  // the two sums that norm's backward needs -- sum dz and sum dz * yhat with yhat = (nx - mean) rstd, dz = dy act'(yhat) (norm.hip,
  // BwdMoments) -- into nsums [N][Ks][mom_slots][2]: the norm's backward then needs no pass of its own over (x, dy) for them.
  const bf16_t* nx;
  const float* nstats;
  float* nsums;
  int nact;
  float nslope;
};

// ABL: timing-only ablation bits (results are wrong when non-zero): 1 = no DMA after the prologue,
// 2 = no barrier, 4 = fragments read once per tap-group only (no per-k-step LDS reads), 8 = no vmcnt waits,
// 16 = every block reads the same weight rows, 32 = every block reads the same input patch (both: L2 hits only)
// SINGLE: one patch buffer instead of two -- for 64-channel inputs (one slab per tile, nothing to prefetch).
// With the 64-wide N tile the block then needs 74 KiB of LDS and TWO blocks share a CU, overlapping one
// block's patch load / epilogue with the other's MFMAs (these K = 576 layers are prologue-bound).
// MF16: v_mfma_f32_16x16x32_bf16 instead of 32x32x16 -- same LDS bytes and MFMA cycles per FLOP, but the chip holds
// a higher clock on it (MI355X_MICROARCH.md, DVFS item 7).  The fragment rows are then 16 pixels x 4 k-chunks, which
// needs the swizzle chunk ^ (row & 6) instead of chunk ^ ((row >> 1) & 7) to stay conflict-free at any tap shift.
// VIRT: the data gradient of a REFLECT-padded 3x3 conv in one launch.  The adjoint of the padding adds the padded-domain
// gradient of row -1 into row 1, of row H into row H-2 (columns alike).  Row -1 of that gradient is the u = 2 taps applied to dy
// row 0 -- the same taps that output row 1 applies to dy row 2 -- so output row 1 reads, for those taps only, the FRAME row
// dy[0] + dy[2] instead of dy[2] (bottom: output row H-2, taps u = 0, dy[H-3] + dy[H-1]; columns alike; the corner pixel holds
// the four-term sum).  The frame (ring_frame_kernel, 2 (W + 2) + 2 H pixels per image) is loaded into the patch positions
// that hold the zero padding otherwise; the reads that must still see that padding (output row 0 at u = 0, ...) go to a zero
// pixel in the slack of the patch buffer.  Costs an add and a min per fragment address and tap; replaces the four ring-strip
// GEMMs + ring_fold_kernel.  Needs py = px = 1, IH = OH, IW = OW, OH >= 8.
template <int TH, int TN, int ABL = 0, bool SINGLE = false, bool MF16 = false, bool MOM = false, bool VIRT = false, bool NSUM = false>
__global__ __launch_bounds__(512) void gemm_halo_kernel(const HaloArgs a) {
  constexpr int R = 3, S = 3, TAPS = 9;
  constexpr int NW = 8, WN = 2;                       // waves: TH (=4) x 2
  static_assert(TH == 4, "8 waves = 4 image rows x 2 channel halves");
  constexpr int BN = WN * TN * 32;
  constexpr int PH = TH + R - 1, PW = 64 + S - 1, NP = PH * PW;
  constexpr int UH = (NP + 7) / 8;                    // 1 KiB DMA units of the patch: 8 pixels x 128 B (64 channels)
  constexpr int HALO = UH * 1024;
  constexpr int B_STAGE = BN * 128;
  constexpr int B_UNITS = BN / 8, BU = (B_UNITS + NW - 1) / NW;
  constexpr int HU = (UH + NW - 1) / NW;              // patch units per wave (7 for 50 units / 8 waves)
  static_assert(HU <= TAPS - 2, "patch units of the next slab are spread over taps 0..HU-1");
  constexpr int PZ = UH * 8 - 1;                      // VIRT: the last slack pixel of a patch buffer is kept zero
  static_assert(!VIRT || (PZ - 1 >= NP && (PZ & 1) == 1 && !MF16 && !MOM), "VIRT needs two slack pixels (even, odd); written for the plain 32x32 form");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo0 = smem;
  constexpr int NBUF = SINGLE ? 1 : 2;
  char* const bring = smem + NBUF * HALO;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_w = a.OW / 64, tiles_h = a.OH / TH;
  const int tiles_m = a.N * tiles_h * tiles_w;
  // XCD-aware tile order (blocks are dealt round-robin to the 8 XCDs, each with its own L2)
  int bid = blockIdx.x;
  const int nblk = gridDim.x;
  if (a.xcd_mode != 0 && (nblk & 7) == 0) bid = (bid & 7) * (nblk >> 3) + (bid >> 3);
  int tile_m = bid % tiles_m, tile_n = bid / tiles_m;
  if (a.xcd_mode == 2 && (tiles_m & 1) == 0) {
    // pairs of N-tiles share half of the patches: (tile_n pair, half) <- old tile_n; inside: patch fastest
    const int tiles_n = nblk / tiles_m;
    if ((tiles_n & 1) == 0) {
      const int hm = tiles_m >> 1;
      const int grp = bid / tiles_m, within = bid - grp * tiles_m;      // grp = old tile_n
      const int pair = grp >> 1, half = grp & 1;
      tile_n = 2 * pair + (within / hm);
      tile_m = half * hm + (within % hm);
    }
  }
  if (a.xcd_mode == 3 && nblk == 256 && tiles_m == 32) {
    // one round of 256 tiles = 32 patches x 8 channel tiles: XCD x = (patch quarter x >> 1, channel half x & 1) takes 8 patches x 4
    // channel tiles -- the split that minimises what the 8 L2s pull through the fabric together (9.4 MB of filter + 6.5 MB of
    // patches each: 127 MB per launch instead of 177 MB with block b -> tile b, which gives every XCD the whole filter)
    const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    tile_n = (x & 1) * 4 + (j >> 3);
    tile_m = (x >> 1) * 8 + (j & 7);
  }
  const int tw_i = tile_m % tiles_w, t1 = tile_m / tiles_w;
  const int th_i = t1 % tiles_h, n = t1 / tiles_h;
  const int oh0 = th_i * TH, ow0 = tw_i * 64, n0 = tile_n * BN;
  const int lrow = lane >> 3, lslot = lane & 7;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);

  // ---- weight tile DMA units
  const bf16_t* b_ptr[BU];
  int b_lds[BU];
  bool b_on[BU];
  const long long ktot = (long long)TAPS * a.Cs;
#pragma unroll
  for (int j = 0; j < BU; ++j) {
    const int u = wid + j * NW;
    b_on[j] = u < B_UNITS;
    const int uu = b_on[j] ? u : 0;
    const int row = uu * 8 + lrow;
    int br = ((ABL & 16) ? 0 : n0) + row;       // ABL 16: every block streams the SAME weight rows (L2-resident)
    br = br < a.b_rows ? br : a.b_rows - 1;
    b_ptr[j] = a.B + (long long)br * ktot + (MF16 ? (lslot ^ (row & 6)) : ((lslot ^ (row >> 1)) & 7)) * 8;
    b_lds[j] = uu * 1024;
  }
  // Prologue order (round 4): weight tile 0 goes on the wire FIRST, each patch unit of slab 0 as soon as its address is known, weight
  // tile 1 last -- the ~1,000 instructions of address set-up below used to run before the first DMA was issued, with the fill latency
  // exposed behind them on every tile.  The first wait of the loop (at most BU pieces outstanding) still means "patch and tile 0 have
  // landed, tile 1 may be in flight".
#pragma unroll
  for (int j = 0; j < BU; ++j)
    if (b_on[j]) glds16(b_ptr[j], bring + b_lds[j]);                                // tile 0: slab 0, tap 0 -> stage 0
  // ---- patch DMA units of this wave: element offset of the lane's 16 bytes (slab 0), or -1 -> zero page
  long long h_off[HU];
  int h_lds[HU];
  int n_hu = 0;
#pragma unroll
  for (int i = 0; i < HU; ++i) {
    const int u = wid + i * NW;
    const bool on = u < UH;
    n_hu += on ? 1 : 0;
    const int ug = on ? u : 0;
    int p = ug * 8 + lrow;
    const int swz_p = p;                              // the swizzle uses the LDS pixel index, also for clamped lanes
    const bool slack = p >= NP;
    p = p < NP ? p : NP - 1;
    const int hr = p / PW, wc = p - hr * PW;
    int ih = ((ABL & 32) ? 0 : oh0) - a.py + hr, iw = ((ABL & 32) ? 0 : ow0) - a.px + wc;   // ABL 32: same patch
    bool ok = true;
    if (a.reflect) {
      ih = ih < 0 ? -ih : (ih >= a.IH ? 2 * (a.IH - 1) - ih : ih);
      iw = iw < 0 ? -iw : (iw >= a.IW ? 2 * (a.IW - 1) - iw : iw);
    } else {
      ok = ((unsigned)ih < (unsigned)a.IH) & ((unsigned)iw < (unsigned)a.IW);
    }
    const int chunk = MF16 ? (lslot ^ (swz_p & 6)) : ((lslot ^ (swz_p >> 1)) & 7);
    h_off[i] = ok ? (((long long)((ABL & 32) ? 0 : n) * a.IH + ih) * a.IW + iw) * a.Cs + chunk * 8 : -1;
    if constexpr (VIRT) {
      // padding pixels come from the frame: rows [2][IW + 2] (column index iw + 1), then columns [2][IH]; slack pixels are zeros
      if (!ok && !slack) {
        const int fw = a.IW + 2;
        // A frame COLUMN pixel has exactly one reader: the lane whose pixel is redirected to it (w == 1 at tap column 2 for the
        // left frame column, w == 62 at tap column 0 for the right one).  A 16-lane group of ds_read_b128 covers the 64 banks exactly
        // once when lane pairs read (even, odd) pixels of one swizzle class, so a redirected lane must land on a pixel of the SAME
        // PARITY AND CLASS as the one it reads otherwise (patch columns 3 / 62): the frame pixel is stored with that class, and when
        // the tile spans the image width (both edges: the 64-pixel-wide ResnetBlock images) the two frame columns swap places -- the
        // right one goes to patch column 0 (even, like 62), the left one to column 65 (odd, like 3).  (Round 4, VERDICT r3 item 8:
        // 14 % of the LDS cycles of this instantiation were bank conflicts.)
        const bool both = ow0 == 0 && ow0 + 64 == a.IW;
        const bool colf = iw < 0 || iw >= a.IW;
        const int iws = (both && colf) ? (iw < 0 ? a.IW : -1) : iw;
        const int fp = (ih < 0 || ih >= a.IH) ? (ih < 0 ? 0 : fw) + iws + 1 : 2 * fw + (iws < 0 ? 0 : a.IH) + ih;
        const int cls = iws < 0 ? hr * PW + 3 : (iws >= a.IW ? hr * PW + 62 : swz_p);
        const int chunk_v = (lslot ^ (cls >> 1)) & 7;
        h_off[i] = -2 - (((long long)n * (2 * fw + 2 * a.IH) + fp) * a.Cs + chunk_v * 8);
      }
      if (slack) h_off[i] = -1;
    }
    h_lds[i] = ug * 1024;
    if (on) {                                           // = issue_patch_unit(i, 0)
      const bf16_t* src = h_off[i] >= 0 ? a.X + h_off[i] : zero;
      if constexpr (VIRT) src = h_off[i] < -1 ? a.V + (-2 - h_off[i]) : src;
      glds16(src, halo0 + h_lds[i]);
    }
  }
#pragma unroll
  for (int j = 0; j < BU; ++j)
    if (b_on[j]) glds16(b_ptr[j] + a.Cs, bring + B_STAGE + b_lds[j]);               // tile 1: slab 0, tap 1 -> stage 1
  int n_b = 0;
#pragma unroll
  for (int j = 0; j < BU; ++j) n_b += b_on[j] ? 1 : 0;

  // ---- fragment addressing
  constexpr int FM = MF16 ? 4 : 2, FN = MF16 ? 2 * TN : TN;   // fragment blocks per wave tile (64 px x TN*32 channels)
  constexpr int FR = MF16 ? 16 : 32;                            // rows per fragment block
  constexpr int KS = MF16 ? 2 : 4;                              // k-steps per 64-channel tile
  int pb[FM];                                         // patch pixel of this lane's row for tap (0,0), per row block
#pragma unroll
  for (int i = 0; i < FM; ++i) pb[i] = wm * PW + i * FR + (lane & (FR - 1));
  // VIRT: column of the patch read by this lane's pixel for tap columns 0 and 2 (>= 1000: the zero pixel), per row block
  int vc0[FM], vc2[FM];
  bool v_top = false, v_bot = false;
  if constexpr (VIRT) {
    v_top = oh0 == 0;
    v_bot = oh0 + TH == a.OH;
    const bool left = ow0 == 0, right = ow0 + 64 == a.OW;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int w = i * FR + (lane & (FR - 1));
      vc0[i] = (left && w == 0) ? 1000 : ((right && w == 62) ? (left ? 0 : 65) : w);       // both edges: frame columns swapped (loader)
      vc2[i] = (right && w == 63) ? 1000 : ((left && w == 1) ? (right ? 65 : 0) : w + 2);
    }
  }
  const int hsel = MF16 ? (lane >> 4) : (lane >> 5);  // k-chunk of this lane inside a k-step
  int b_rd[FN][KS];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int row = wn * TN * 32 + j * FR + (lane & (FR - 1));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      b_rd[j][ks] = MF16 ? (row << 7) + (((4 * ks + hsel) ^ (row & 6)) << 4) : swz128(row, 2 * ks + hsel);
  }

  typedef typename std::conditional<MF16, f32x4, f32x16>::type acc_t;
  acc_t acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < (MF16 ? 4 : 16); ++e) acc[i][j][e] = 0.f;

  const int CC = a.Cs >> 6;
  const int T_total = CC * TAPS;

  auto issue_patch_unit = [&](int i, int slab) {      // i: compile-time after unrolling at the call sites
    char* const dst = halo0 + (slab & (NBUF - 1)) * HALO + h_lds[i];
    const bf16_t* src = h_off[i] >= 0 ? a.X + h_off[i] + slab * 64 : zero;
    if constexpr (VIRT) src = h_off[i] < -1 ? a.V + (-2 - h_off[i]) + slab * 64 : src;
    glds16(src, dst);
  };
  auto issue_b = [&](int tile) {                      // tile = slab*9 + tap -> K offset tap*Cs + slab*64
    const int slab = tile / TAPS, tap = tile - slab * TAPS;
    char* const st = bring + (tile % 3) * B_STAGE;
    const long long koff = (long long)tap * a.Cs + slab * 64;
#pragma unroll
    for (int j = 0; j < BU; ++j)
      if (b_on[j]) glds16(b_ptr[j] + koff, st + b_lds[j]);
  };

  // (the prologue DMA -- weight tile 0, the patch of slab 0, weight tile 1 -- was issued above, between the address computations)

  // Main loop: slabs outside, the 9 taps unrolled -- the tap offset, the ring stage (tap % 3), the patch unit to
  // prefetch and every end-of-range test but "last slab?" are compile-time constants.  (With a runtime tap the
  // selection of the patch unit and of the vmcnt count compiled to ~20 scalar branches per tap, sitting right after
  // the barrier where all 8 waves wait for them.)
  // In flight across the barrier: only weight tile t+1 (BU pieces per wave, issued last in iteration t-1); the patch
  // unit issued before it is one iteration old by then, so a constant count suffices.
  static_assert(B_UNITS % NW == 0, "every wave issues exactly BU weight pieces per tile");
  auto tap_body = [&](const int tap_r, const int tap_s, const int slab, const bool more, const char* const hb) {
    const int tap = tap_r * S + tap_s;               // tap_s is a compile-time constant in both callers
    if (!(ABL & 8)) {
      if (tap < TAPS - 1 || more) wait_vmcnt<BU>(); else wait_vmcnt<0>();
    }
    if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
    // issue group t+2: one patch unit of the NEXT slab (from the slab's first tap on: its buffer was being
    // read until the previous slab ended), then weight tile t+2
    auto issue_group = [&]() {
      if (!(ABL & 1)) {
        if constexpr (!SINGLE) {
          if (tap < HU && tap < n_hu && more) {
#pragma unroll
            for (int i = 0; i < HU; ++i)
              if (i == tap) issue_patch_unit(i, slab + 1);          // tap is a constant here (unrolled caller)
          }
        }
        const int tap2 = tap + 2 < TAPS ? tap + 2 : tap + 2 - TAPS;
        if (tap + 2 < TAPS || more) {
          char* const st2 = bring + ((tap_s + 2) % 3) * B_STAGE;    // (t + 2) % 3 with t = 9 * slab + 3 * tap_r + tap_s
          const long long koff = (long long)tap2 * a.Cs + (slab + (tap + 2 < TAPS ? 0 : 1)) * 64;
#pragma unroll
          for (int jj = 0; jj < BU; ++jj) glds16(b_ptr[jj] + koff, st2 + b_lds[jj]);
        }
      }
    };
    // (Tried and removed in round 4, negative results on record in DESIGN.md 4.1: waves 4..7 issuing their DMA group AFTER the MFMA
    // cluster -- the stagger of MI355X_MICROARCH.md "Two waves per SIMD" item 9 -- cost 8 %, round-2 developer mode 23; fragment
    // reads as a hand-written software pipeline with counted lgkmcnt were neutral, mode 25: hipcc already interleaves these plain
    // ds_read_b128.)
    const char* const st = bring + tap_s * B_STAGE;    // tap % 3
    const int tapoff = tap_r * PW + tap_s;
    int a_base[FM], a_sw[FM];
    int vrow = 0;
    if constexpr (VIRT) {
      // patch row read by this wave's image row for filter row tap_r: redirected to the frame row / the zero pixel at the edges
      vrow = (wm + tap_r) * PW;
      if (tap_r == 2) vrow = (v_top && wm == 1) ? 0 : ((v_bot && wm == 3) ? 1000 : vrow);
      if (tap_r == 0) vrow = (v_bot && wm == 2) ? 5 * PW : ((v_top && wm == 0) ? 1000 : vrow);
    }
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      int pt = pb[i] + tapoff;
      int psw = pt;                                   // pixel whose swizzle class the read uses
      if constexpr (VIRT) {
        const int ncol = i * FR + (lane & (FR - 1)) + tap_s;      // the column this lane reads without a redirect
        const int col = tap_s == 0 ? vc0[i] : (tap_s == 2 ? vc2[i] : ncol);
        const int ptv = vrow + col;
        // zero pixel: every chunk of it is zero, so the lane keeps the class of its un-redirected pixel; frame column: stored
        // with the class of (patch row, ncol) by the loader above.  Either way the lane's bank is the one of the regular pattern.
        psw = ptv >= PZ ? pt : vrow + ncol;
        pt = ptv < PZ ? ptv : PZ - 1 + (pt & 1);      // two zero pixels (PZ - 1 even, PZ odd): the parity of the un-redirected pixel
      }
      a_base[i] = pt << 7;
      a_sw[i] = MF16 ? (psw & 6) << 4 : ((psw >> 1) & 7) << 4;
    }
    issue_group();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s16x8 af[FM], bf[FN];
      const int ks_r = (ABL & 4) ? 0 : ks;
#pragma unroll
      for (int i = 0; i < FM; ++i)
        af[i] = *reinterpret_cast<const s16x8*>(hb + a_base[i] + ((((MF16 ? 4 : 2) * ks_r + hsel) << 4) ^ a_sw[i]));
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const s16x8*>(st + b_rd[j][ks_r]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          if constexpr (MF16) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  if constexpr (SINGLE) {
    // one slab (Cs == 64), nothing to prefetch: the rolled loop keeps the kernel within 128 VGPRs (two blocks per CU)
    // (filter rows rolled, the 3 taps of a row unrolled: ring stage and tap column stay compile-time constants)
#pragma unroll 1
    for (int r = 0; r < R; ++r) {
#pragma unroll
      for (int sx = 0; sx < S; ++sx) tap_body(r, sx, 0, false, halo0);
    }
  } else {
    for (int slab = 0; slab < CC; ++slab) {
      const bool more = slab + 1 < CC;                  // wave-uniform
      const char* const hb = halo0 + (slab & (NBUF - 1)) * HALO;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) tap_body(tap / S, tap % S, slab, more, hb);
    }
  }

  // ---- optional moments of the block's 4 x 64 pixels per channel for the InstanceNorm that follows (ResnetBlock convs):
  // (mean, M2) of the bf16-rounded values about a per-lane pilot, lane halves merged, then the four row-waves through LDS
  // with Chan's formula in a fixed order (common.h) -- the norm's own moment kernel (a launch per norm) is not needed
  if constexpr (MOM && !MF16) {
    static_assert(FM == 2, "moments are written for the 32x32 accumulator layout");
    __syncthreads();                                    // the pipeline LDS is dead
    float* const red = reinterpret_cast<float*>(smem);  // [TH][BN][2]
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const float pilot = bf16_round(acc[0][j][0]);
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = bf16_round(acc[i][j][e]) - pilot;
          s1 += v;
          s2 += v * v;
        }
      float mean, m2;                                   // lanes l and l + 32 hold the same column: 32 of the row's 64 pixels each
      shifted_to_mean_m2(s1, s2, pilot, 32.f, mean, m2);
      const float mean_o = __shfl_xor(mean, 32, 64), m2_o = __shfl_xor(m2, 32, 64);
      chan_merge_equal(mean, m2, mean_o, m2_o, 32.f);
      if (lane < 32) {
        const int col = wn * TN * 32 + j * 32 + lane;
        red[(wm * BN + col) * 2] = mean;
        red[(wm * BN + col) * 2 + 1] = m2;
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.Ks) {
      float msum = 0.f;
#pragma unroll
      for (int r = 0; r < TH; ++r) msum += red[(r * BN + tid) * 2];
      const float mean = msum * (1.f / TH);
      float m2 = 0.f;
#pragma unroll
      for (int r = 0; r < TH; ++r) {
        const float dm = red[(r * BN + tid) * 2] - mean;
        m2 += red[(r * BN + tid) * 2 + 1] + 64.f * dm * dm;
      }
      float* const o = a.mom + (((long long)n * a.Ks + n0 + tid) * a.mom_slots + (th_i * tiles_w + tw_i)) * 2;
      o[0] = mean;
      o[1] = m2;
    }
  }
  // ---- epilogue: wave wm owns image row oh0+wm (2 x 32 consecutive pixels); staged through LDS so that the
  // global stores are 16-byte channel vectors (see acc_tile_to_lds)
  __syncthreads();
  constexpr int PITCH = BN * 2 + 64;
  static_assert(TH * 64 * PITCH <= NBUF * HALO + 3 * B_STAGE, "epilogue tile fits the pipeline LDS");
  if constexpr (MF16) acc16_tile_to_lds<FM, FN>(smem, PITCH, wm * 64, wn * TN * 32, n0, lane, acc, a.bias, a.Kout, a.act, a.slope);
  else acc_tile_to_lds<2, TN>(smem, PITCH, wm * 64, wn * TN * 32, n0, lane, acc, a.bias, a.Kout, a.act, a.slope);
  __syncthreads();
  const long long blk_base = a.out_base + n * a.out_sn + (long long)oh0 * a.out_sh + (long long)ow0 * a.out_sw;
  constexpr int VPR = BN / 8;
  // The fused operands (fan-in addend, ReLU mask) are loaded for ALL of the thread's vectors before the first store: written as
  // one loop, each load sat behind the previous store (the compiler cannot rule out that Y aliases them) and the epilogue
  // paid one memory round trip per vector -- 8 in a row with the block alone on its CU.
  constexpr int NV = TH * 64 * VPR / 512;
  static_assert(TH * 64 * VPR % 512 == 0, "every thread owns the same number of output vectors");
  if constexpr (NSUM) {
    // Same epilogue as the fused form below, plus the sums of the InstanceNorm backward that consumes Y (HaloArgs::nx).  A
    // thread's vectors all lie in ONE 8-channel column (512 % VPR == 0), so its sums stay in registers over its NV pixels; the
    // four lanes of a wave that share a column are merged by two xor shuffles, the eight waves through LDS in wave order
    // (fixed order: deterministic), one slot per block and channel.
    static_assert(512 % VPR == 0 && !MF16 && !SINGLE, "one channel column per thread");
    constexpr int NRED_OFF = TH * 64 * PITCH;
    static_assert(NRED_OFF + NW * BN * 2 * 4 <= NBUF * HALO + 3 * B_STAGE, "the reduction rows fit behind the epilogue tile");
    const int v = tid % VPR;
    const bool on = n0 + v * 8 < a.Ks;
    u32x4 addv[NV], mskv[NV], nxv[NV];
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int row = (tid + 512 * it) / VPR;
      const long long off = blk_base + (long long)(row >> 6) * a.out_sh + (long long)(row & 63) * a.out_sw + n0 + v * 8;
      if (on && a.addend != nullptr) addv[it] = *reinterpret_cast<const u32x4*>(a.addend + off);
      if (on && a.mask != nullptr) mskv[it] = *reinterpret_cast<const u32x4*>(a.mask + off);
      if (on) nxv[it] = *reinterpret_cast<const u32x4*>(a.nx + off);
    }
    float mean[8], rstd[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { mean[e] = 0.f; rstd[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
    if (on) {
      const f32x4* const st = reinterpret_cast<const f32x4*>(a.nstats + ((long long)n * a.Ks + n0 + v * 8) * 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 t = st[q];
        mean[2 * q] = t[0]; rstd[2 * q] = t[1]; mean[2 * q + 1] = t[2]; rstd[2 * q + 1] = t[3];
      }
    }
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int row = (tid + 512 * it) / VPR;
      if (!on) continue;
      const long long off = blk_base + (long long)(row >> 6) * a.out_sh + (long long)(row & 63) * a.out_sw + n0 + v * 8;
      u32x4 val = *reinterpret_cast<const u32x4*>(smem + row * PITCH + v * 16);
      if (a.addend != nullptr) val = add_bf16x8(val, addv[it]);
      if (a.mask != nullptr) val = relu_mask8(val, mskv[it]);
      *reinterpret_cast<u32x4*>(a.Y + off) = val;
      float g[8], x[8];
      Vec16<bf16_t>::unpack(val, g);                   // the ROUNDED gradient: what the norm's backward re-reads
      Vec16<bf16_t>::unpack(nxv[it], x);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float yh = (x[e] - mean[e]) * rstd[e];
        float dz = g[e];
        if (a.nact == JPDSE_ACT_RELU) dz = yh > 0.f ? dz : 0.f;
        else if (a.nact == JPDSE_ACT_LRELU) dz = yh > 0.f ? dz : dz * a.nslope;
        s1[e] += dz;
        s2[e] += dz * yh;
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] += __shfl_xor(s1[e], 16, 64);
      s2[e] += __shfl_xor(s2[e], 16, 64);
      s1[e] += __shfl_xor(s1[e], 32, 64);
      s2[e] += __shfl_xor(s2[e], 32, 64);
    }
    float* const nred = reinterpret_cast<float*>(smem + NRED_OFF);     // [NW][BN][2], behind the tile the other waves still read
    if (lane < VPR) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        nred[(wid * BN + v * 8 + e) * 2] = s1[e];
        nred[(wid * BN + v * 8 + e) * 2 + 1] = s2[e];
      }
    }
    __syncthreads();
    if (tid < BN && n0 + tid < a.Ks) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        t1 += nred[(w * BN + tid) * 2];
        t2 += nred[(w * BN + tid) * 2 + 1];
      }
      float* const o = a.nsums + (((long long)n * a.Ks + n0 + tid) * a.mom_slots + (th_i * tiles_w + tw_i)) * 2;
      o[0] = t1;
      o[1] = t2;
    }
    return;
  }
  if (a.addend != nullptr || a.mask != nullptr) {
    u32x4 addv[NV], mskv[NV];
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int idx = tid + 512 * it;
      const int row = idx / VPR, v = idx - row * VPR;
      const long long off = blk_base + (long long)(row >> 6) * a.out_sh + (long long)(row & 63) * a.out_sw + n0 + v * 8;
      const bool on = n0 + v * 8 < a.Ks;
      if (on && a.addend != nullptr) addv[it] = *reinterpret_cast<const u32x4*>(a.addend + off);
      if (on && a.mask != nullptr) mskv[it] = *reinterpret_cast<const u32x4*>(a.mask + off);
    }
#pragma unroll
    for (int it = 0; it < NV; ++it) {
      const int idx = tid + 512 * it;
      const int row = idx / VPR, v = idx - row * VPR;
      if (n0 + v * 8 >= a.Ks) continue;
      const long long off = blk_base + (long long)(row >> 6) * a.out_sh + (long long)(row & 63) * a.out_sw + n0 + v * 8;
      u32x4 val = *reinterpret_cast<const u32x4*>(smem + row * PITCH + v * 16);
      if (a.addend != nullptr) val = add_bf16x8(val, addv[it]);
      if (a.mask != nullptr) val = relu_mask8(val, mskv[it]);
      *reinterpret_cast<u32x4*>(a.Y + off) = val;
    }
    return;
  }
  for (int idx = tid; idx < TH * 64 * VPR; idx += 512) {
    const int row = idx / VPR, v = idx - row * VPR;
    if (n0 + v * 8 >= a.Ks) continue;
    const long long off = blk_base + (long long)(row >> 6) * a.out_sh + (long long)(row & 63) * a.out_sw;
    *reinterpret_cast<u32x4*>(a.Y + off + n0 + v * 8) = *reinterpret_cast<const u32x4*>(smem + row * PITCH + v * 16);
  }
  if (a.pool != nullptr) {
    // the 2 x 32 pooled pixels of this 4 x 64 patch, from the tile that is still in LDS: the separate pool pass re-read Y
    const int OH2 = a.OH >> 1, OW2 = a.OW >> 1;
    for (int idx = tid; idx < 2 * 32 * VPR; idx += 512) {
      const int pp = idx / VPR, v = idx - pp * VPR;
      if (n0 + v * 8 >= a.Ks) continue;
      const int pr = pp >> 5, pc = pp & 31;
      const char* const src = smem + ((2 * pr) * 64 + 2 * pc) * PITCH + v * 16;
      const u32x4 q0 = *reinterpret_cast<const u32x4*>(src), q1 = *reinterpret_cast<const u32x4*>(src + PITCH);
      const u32x4 q2 = *reinterpret_cast<const u32x4*>(src + 64 * PITCH), q3 = *reinterpret_cast<const u32x4*>(src + 65 * PITCH);
      *reinterpret_cast<u32x4*>(a.pool + (((long long)n * OH2 + (oh0 >> 1) + pr) * OW2 + (ow0 >> 1) + pc) * a.Ks + n0 + v * 8) =
          max_bf16x8(max_bf16x8(q0, q1), max_bf16x8(q2, q3));
    }
  }
}

}  // namespace jpdse
