// Input-stationary implicit GEMM with a TAP PROGRAM: the generalisation of gemm_halo.h (3x3 stride 1, one accumulator
// set) to the layers that were left on the tap-re-staging gemm_fast_kernel:
//
//   * the data gradient of a stride-2 conv / the forward of a ConvTranspose2d (networks.py:215,244): the four sub-pixel
//     phases of dx are four stride-1 convs over the SAME dy pixels with 4 + 2 + 2 + 1 taps (3x3) or 4 x 4 taps (4x4).
//     gemm_fast ran them as four independent problems: every phase re-stages dy once per tap (9 activation tiles of 32 KiB
//     per 64-channel slab and 256 dy pixels) and the short K loops (1 .. 4 taps) are prologue / epilogue bound.  Here a block
//     stages the (TH+1) x (TW+1) dy patch ONCE per slab and runs a program of taps over it, each tap feeding ONE OF TWO
//     accumulator sets (= two phases): {phase (0,0): 4 taps, phase (1,1): 1 tap} and {phase (0,1): 2, phase (1,0): 2} are the
//     two programs of a 3x3 layer, launched together (block-uniform branch); 4x4 stride 2: {4, 4} twice.
//   * 4x4 stride-1 convs (PatchGAN layer 3, networks.py:430-449) and their data gradient: one set, 16 taps, the patch
//     staged once per slab instead of 16 activation tiles.
//
// Per slab and tap the block streams only the BN x 64 weight tile (3-stage ring, counted vmcnt, ONE raw barrier per tap,
// exactly the loop of gemm_halo.h); LDS-DMA instructions per wave and 16 MFMAs: ~3 instead of gemm_fast's 6.
//
// Geometry (template): 8 waves = 4 row groups x 2 channel halves; WROWS = 1: tile 4 rows x 64 pixels (a wave owns one row,
// two 32-pixel fragment blocks side by side); WROWS = 2: tile 8 rows x 32 pixels (a wave owns two rows, one block each) --
// the smaller patch of the second form is what lets a 4x4 filter's halo be double buffered.  PH x PW = patch size.
// Everything that selects registers or LDS stages is a compile-time constant of the unrolled (slab % 3, tap) body (the
// lesson of gemm_halo.h: scalar control flow behind the barrier idles the matrix pipes of the whole CU); tap offsets and
// panel offsets are kernel arguments (SGPRs).
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "gemm_halo.h"

namespace jpdse {

constexpr int kTapsMax = 16;

struct TapsProg {
  const bf16_t* B[2];        // weight panel of each accumulator set: [b_rows][ktot] with k = tap_koff + channel
  long long ktot[2];         // panel row stride, elements
  long long out_base[2];     // output offset of each set (the phase's first pixel), elements
  int tap_off[kTapsMax];     // patch pixel offset dr * PW + dc of every tap (set 0's taps first)
  int tap_koff[kTapsMax];    // K offset of the tap inside its set's panel (channel slab added by the kernel)
};

struct TapsArgs {
  const bf16_t* X;           // [N][IH][IW][Cs]
  bf16_t* Y;
  const float* bias;         // optional (forward of a biased conv), applied to every set
  int N, OH, OW;             // tile grid of ONE set: OH % TH == 0, OW % TW == 0
  int IH, IW, Cs;
  int py, px;                // patch origin = (oh0 - py, ow0 - px); outside the image: zeros
  int Kout, Ks, b_rows;
  long long out_sn, out_sh, out_sw;
  int act;
  float slope;
  const bf16_t* addend;      // optional, Y's addressing: Y = result + addend (gradient fan-in), then
  const bf16_t* mask;        // optional, Y's addressing: zeroed where mask <= 0 (ReLU backward of the conv's input)
  float mask_slope;          // LeakyReLU backward instead: scaled by this where mask <= 0 (0 = ReLU)
  int reflect;               // 1: patch pixels outside the image are mirrored (ReflectionPad2d), 0: zeros
  // split-K over the channel slabs (one program only; few tiles, long reductions: the 1024-channel ResnetBlocks of the
  // LocalEnhancer trunk at 16 x 32 pixels): block (tile, split) runs slabs [CC split / splits, CC (split + 1) / splits) and stores
  // its fp32 tile into slab `split` of `partial` ([splits][N OH OW][Ks]); splitk_finish_kernel (gemm_fast.h) sums the slabs in
  // index order and applies bias / activation / addend / mask
  int splits;
  float* partial;
  const bf16_t* V;           // VIRT instantiation: the folded frame of X (ring_frame_kernel): reflect data gradient in one launch (gemm_halo.h)
  int nblk0;                 // blocks [0, nblk0) run prog[0], the rest prog[1]
  TapsProg prog[2];
};

// VIRT (3x3 program, T0 = 9 taps in row-major order, py = px = 1, equal grids): the reflect ring folded into a frame of the
// input, exactly as in gemm_halo.h -- the loader puts the frame where the zero padding would be, the fragment addresses of the
// image rows 1 / OH-2 and columns 1 / OW-2 are redirected to it for the filter rows / columns that cross the border, the reads
// that must still see the padding go to a zero pixel in the slack of the patch buffer.
template <int TN, int T0, int T1, int WROWS, int PH, int PW, bool VIRT = false>
__device__ __forceinline__ void gemm_taps_body(const TapsArgs& a, const TapsProg& pr, int bid, char* smem) {
  // (bid is reduced to the tile index below when the launch is split over channel slabs)
  constexpr int NT = T0 + T1, NSETS = T1 > 0 ? 2 : 1;
  constexpr int NW = 8, WN = 2;
  constexpr int TH = 4 * WROWS, TW = 64 / WROWS;
  constexpr int BN = WN * TN * 32;
  constexpr int NP = PH * PW;
  constexpr int UH = (NP + 7) / 8;                    // 1 KiB DMA units of the patch: 8 pixels x 128 B
  constexpr int HALO = UH * 1024;
  constexpr int B_STAGE = BN * 128;
  constexpr int B_UNITS = BN / 8, BU = B_UNITS / NW;
  static_assert(B_UNITS % NW == 0, "every wave issues exactly BU weight pieces per tile");
  constexpr int HU = (UH + NW - 1) / NW;              // patch units per wave
  constexpr int UPT = (HU + NT - 1) / NT;             // patch units of the NEXT slab issued per tap
  static_assert(NT <= kTapsMax && NT >= 2, "tap program length");
  constexpr int PZ = UH * 8 - 1;                      // VIRT: the last slack pixel of a patch buffer is kept zero
  static_assert(!VIRT || (T0 == 9 && T1 == 0 && PZ - 1 >= NP && (PZ & 1) == 1 && (PW & 1) == 0 && PH == TH + 2 && PW == TW + 2), "VIRT: nine taps, one set, two slack pixels");
  // LDS: weight ring first (its stage offsets then fit the 16-bit immediate of ds_read), the two patch buffers behind it
  char* const bring = smem;
  char* const halo0 = smem + 3 * B_STAGE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_w = a.OW / TW, tiles_h = a.OH / TH;
  const int tiles_m = a.N * tiles_h * tiles_w;
  const int ntiles = tiles_m * ((a.Ks + BN - 1) / BN);
  const int split = a.splits > 1 ? bid / ntiles : 0;
  if (a.splits > 1) bid -= split * ntiles;
  const int tile_m = bid % tiles_m, tile_n = bid / tiles_m;
  const int tw_i = tile_m % tiles_w, t1 = tile_m / tiles_w;
  const int th_i = t1 % tiles_h, n = t1 / tiles_h;
  const int oh0 = th_i * TH, ow0 = tw_i * TW, n0 = tile_n * BN;
  const int lrow = lane >> 3, lslot = lane & 7;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);

  // ---- patch DMA units of this wave: element offset of the lane's 16 bytes (slab 0), or -1 -> zero page
  int h_off[HU];                                      // 32-bit element offsets (the launcher checks the tensor size)
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
    int ih = oh0 - a.py + hr, iw = ow0 - a.px + wc;
    bool ok = ((unsigned)ih < (unsigned)a.IH) & ((unsigned)iw < (unsigned)a.IW);
    if (a.reflect) {
      ih = ih < 0 ? -ih : (ih >= a.IH ? 2 * (a.IH - 1) - ih : ih);
      iw = iw < 0 ? -iw : (iw >= a.IW ? 2 * (a.IW - 1) - iw : iw);
      ok = ((unsigned)ih < (unsigned)a.IH) & ((unsigned)iw < (unsigned)a.IW);     // patch rows beyond one reflection: not read by any tap
    }
    const int chunk = (lslot ^ (swz_p >> 1)) & 7;
    h_off[i] = ok ? ((n * a.IH + ih) * a.IW + iw) * a.Cs + chunk * 8 : -1;
    if constexpr (VIRT) {
      // padding pixels come from the frame: rows [2][IW + 2] (column index iw + 1), then columns [2][IH]; slack pixels are zeros
      if (!ok && !slack) {
        const int fw = a.IW + 2;
        // frame column pixels take the swizzle class AND pixel parity of the pixel their one reader reads otherwise; tiles that span
        // the image width swap the two frame columns for that (gemm_halo.h, round 4)
        const bool both = ow0 == 0 && ow0 + TW == a.IW;
        const bool colf = iw < 0 || iw >= a.IW;
        const int iws = (both && colf) ? (iw < 0 ? a.IW : -1) : iw;
        const int fp = (ih < 0 || ih >= a.IH) ? (ih < 0 ? 0 : fw) + iws + 1 : 2 * fw + (iws < 0 ? 0 : a.IH) + ih;
        const int cls = iws < 0 ? hr * PW + 3 : (iws >= a.IW ? hr * PW + TW - 2 : swz_p);
        const int chunk_v = (lslot ^ (cls >> 1)) & 7;
        h_off[i] = -2 - ((n * (2 * fw + 2 * a.IH) + fp) * a.Cs + chunk_v * 8);
      }
      if (slack) h_off[i] = -1;
    }
    h_lds[i] = ug * 1024;
  }
  // ---- weight tile DMA units, per accumulator set
  int b_off[NSETS][BU];                               // 32-bit element offsets into each set's panel
  int b_lds[BU];
#pragma unroll
  for (int j = 0; j < BU; ++j) {
    const int u = wid + j * NW;
    const int row = u * 8 + lrow;
    int br = n0 + row;
    br = br < a.b_rows ? br : a.b_rows - 1;
#pragma unroll
    for (int q = 0; q < NSETS; ++q) b_off[q][j] = br * (int)pr.ktot[q] + ((lslot ^ (row >> 1)) & 7) * 8;
    b_lds[j] = u * 1024;
  }

  // (round 4: the prologue DMA is issued HERE, before the fragment addressing and the zeroing of up to 128 accumulator registers, so that the first fill flies under them)
  const int CC_all = a.Cs >> 6;
  const int s_begin = a.splits > 1 ? CC_all * split / a.splits : 0;
  const int CC = a.splits > 1 ? CC_all * (split + 1) / a.splits : CC_all;      // slabs [s_begin, CC)

  auto issue_patch_unit = [&](int i, int slab) {      // i: compile-time after unrolling at the call sites
    char* const dst = halo0 + (slab & 1) * HALO + h_lds[i];
    const bf16_t* src = h_off[i] >= 0 ? a.X + (h_off[i] + slab * 64) : zero;
    if constexpr (VIRT) src = h_off[i] < -1 ? a.V + ((-2 - h_off[i]) + slab * 64) : src;
    glds16(src, dst);
  };
  auto issue_b = [&](const int tap, const int slab, const int stage) {      // tap, stage: compile-time at the call sites
    const int q = tap < T0 ? 0 : 1;
    char* const st = bring + stage * B_STAGE;
    const bf16_t* const base = pr.B[q] + (pr.tap_koff[tap] + slab * 64);      // scalar
#pragma unroll
    for (int j = 0; j < BU; ++j) glds16(base + b_off[q][j], st + b_lds[j]);
  };

  // prologue: whole patch of slab 0, then weight tiles 0 and 1 (NT >= 2)
#pragma unroll
  for (int i = 0; i < HU; ++i)
    if (i < n_hu) issue_patch_unit(i, s_begin);
  issue_b(0, s_begin, 0);
  issue_b(1, s_begin, 1);

  // ---- fragment addressing
  int pb[2];                                          // patch pixel of this lane's row for tap offset 0, per fragment block
#pragma unroll
  for (int i = 0; i < 2; ++i)
    pb[i] = WROWS == 1 ? wm * PW + i * 32 + (lane & 31) : (2 * wm + i) * PW + (lane & 31);
  // VIRT: patch column read by this lane's pixel for tap columns 0 and 2 (>= 1000: the zero pixel); edge flags of the tile
  int vc0 = 0, vc2 = 0;
  bool v_top = false, v_bot = false;
  if constexpr (VIRT) {
    static_assert(!VIRT || WROWS == 2, "VIRT is written for the 8 x 32 tile (one fragment block per image row)");
    v_top = oh0 == 0;
    v_bot = oh0 + TH == a.OH;
    const bool left = ow0 == 0, right = ow0 + TW == a.OW;
    const int w = lane & 31;
    vc0 = (left && w == 0) ? 1000 : ((right && w == TW - 2) ? (left ? 0 : PW - 1) : w);
    vc2 = (right && w == TW - 1) ? 1000 : ((left && w == 1) ? (right ? PW - 1 : 0) : w + 2);
  }
  const int hsel = lane >> 5;                         // k-chunk of this lane inside a k-step
  int b_rd[TN][4];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int row = wn * TN * 32 + j * 32 + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) b_rd[j][ks] = swz128(row, 2 * ks + hsel);
  }

  f32x16 acc[NSETS][2][TN];
#pragma unroll
  for (int q = 0; q < NSETS; ++q)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[q][i][j][e] = 0.f;

  // One (slab, tap) step.  SM = slab % 3 and TAP are compile-time: ring stage (SM * NT + TAP) % 3, accumulator set, the
  // patch units of the next slab to prefetch and the tap two steps ahead are constants; `more` (another slab follows) and
  // `more2` (two more follow) are wave-uniform scalars.
  auto step = [&](const int SM, const int TAP, const int slab, const bool more, const char* const hb) {
    if (TAP < NT - 1 || more) wait_vmcnt<BU>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // issue group: up to UPT patch units of the NEXT slab (its buffer was last read in the previous slab), then the
    // weight tile of step t + 2
    if (more) {
#pragma unroll
      for (int i = 0; i < HU; ++i)
        if (i >= TAP * UPT && i < (TAP + 1) * UPT && i < n_hu) issue_patch_unit(i, slab + 1);
    }
    {
      const int t2 = TAP + 2;
      const int stage2 = (SM * NT + TAP + 2) % 3;
      if (t2 < NT) issue_b(t2, slab, stage2);
      else if (more) issue_b(t2 - NT, slab + 1, stage2);
    }
    const int q = TAP < T0 ? 0 : 1;
    const char* const st = bring + ((SM * NT + TAP) % 3) * B_STAGE;
    // The tap's fragment addresses are formed HERE, every step: left to itself hipcc hoists the address arithmetic of all
    // 3 x NT unrolled steps out of the slab loop (gemm_halo_kernel: 180 VGPRs next to 64 accumulators), which with the 128
    // accumulator registers of two sets spills to scratch.  The empty asm makes the tap offset opaque at this point.
    int tapoff = pr.tap_off[TAP];
    asm volatile("" : "+s"(tapoff));
    int a_base[2], a_sw[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      int pt = pb[i] + tapoff;
      int psw = pt;                                   // pixel whose swizzle class the read uses (gemm_halo.h: redirected lanes keep their bank)
      if constexpr (VIRT) {
        const int tr = TAP / 3, ts = TAP % 3;         // row-major nine-tap program (the launcher sets tap_off = tr * PW + ts); constants after unrolling
        const int j = 2 * wm + i;                     // image row of this fragment block inside the tile
        int vrow = (j + tr) * PW;
        if (tr == 2) vrow = (v_top && j == 1) ? 0 : ((v_bot && j == TH - 1) ? 1000 : vrow);
        if (tr == 0) vrow = (v_bot && j == TH - 2) ? (PH - 1) * PW : ((v_top && j == 0) ? 1000 : vrow);
        const int ncol = (lane & 31) + ts;
        const int col = ts == 0 ? vc0 : (ts == 2 ? vc2 : ncol);
        const int ptv = vrow + col;
        psw = ptv >= PZ ? pt : vrow + ncol;
        pt = ptv < PZ ? ptv : PZ - 1 + (pt & 1);
      }
      a_base[i] = pt << 7;
      a_sw[i] = ((psw >> 1) & 7) << 4;
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      s16x8 af[2], bf[TN];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const s16x8*>(hb + a_base[i] + (((2 * ks + hsel) << 4) ^ a_sw[i]));
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const s16x8*>(st + b_rd[j][ks]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[q][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[q][i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
  };

  // slabs in groups of three (the ring stage of a step repeats every 3 steps; NT is not a multiple of 3 in general)
  for (int slab0 = s_begin; slab0 < CC; slab0 += 3) {
#pragma unroll
    for (int sm = 0; sm < 3; ++sm) {
      const int slab = slab0 + sm;
      if (slab >= CC) break;                            // wave-uniform, once per slab
      const bool more = slab + 1 < CC;
      const char* const hb = halo0 + (slab & 1) * HALO;
#pragma unroll
      for (int tap = 0; tap < NT; ++tap) step(sm, tap, slab, more, hb);
    }
  }

  // ---- split-K: the fp32 accumulators go to this split's slab as they are (row m = the pixel's linear index)
  if (a.splits > 1) {
    float* const slab = a.partial + (long long)split * ((long long)a.N * a.OH * a.OW) * a.Ks;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * TN * 32 + j * 32 + (lane & 31);
      if (col >= a.Ks) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rl = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);      // 0..63: the wave's pixel inside its 64
          const int r = WROWS == 1 ? wm : 2 * wm + (rl >> 5), c = WROWS == 1 ? rl : (rl & 31);
          const long long m = ((long long)n * a.OH + oh0 + r) * a.OW + ow0 + c;
          slab[m * a.Ks + col] = acc[0][i][j][e];
        }
      }
    }
    return;
  }
  // ---- epilogue, one accumulator set at a time through the same LDS tile: 16-byte channel vectors to global
  constexpr int PITCH = BN * 2 + 64;
  static_assert(TH * TW * PITCH <= 2 * HALO + 3 * B_STAGE, "epilogue tile fits the pipeline LDS");
  constexpr int VPR = BN / 8;
#pragma unroll
  for (int q = 0; q < NSETS; ++q) {
    __syncthreads();
    // tile row of this wave's fragment block i: WROWS == 1: wm * 64 + i * 32 (+ lane row); WROWS == 2: (2 wm + i) * 32
    acc_tile_to_lds<2, TN>(smem, PITCH, WROWS == 1 ? wm * 64 : wm * 64, wn * TN * 32, n0, lane, acc[q], a.bias, a.Kout, a.act, a.slope);
    __syncthreads();
    const long long blk_base = pr.out_base[q] + n * a.out_sn + (long long)oh0 * a.out_sh + (long long)ow0 * a.out_sw;
    if (a.addend != nullptr || a.mask != nullptr) {
      // fused operands loaded for all of the thread's vectors before the first store (gemm_halo.h: behind a store the compiler
      // cannot hoist them, and the epilogue would pay one memory round trip per vector)
      constexpr int NV = TH * TW * VPR / 512;
      static_assert(TH * TW * VPR % 512 == 0, "every thread owns the same number of output vectors");
      u32x4 addv[NV], mskv[NV];
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const int idx = tid + 512 * it;
        const int row = idx / VPR, v = idx - row * VPR;
        const int r = row / TW, c = row - r * TW;
        const long long off = blk_base + (long long)r * a.out_sh + (long long)c * a.out_sw + n0 + v * 8;
        const bool on = n0 + v * 8 < a.Ks;
        if (on && a.addend != nullptr) addv[it] = *reinterpret_cast<const u32x4*>(a.addend + off);
        if (on && a.mask != nullptr) mskv[it] = *reinterpret_cast<const u32x4*>(a.mask + off);
      }
#pragma unroll
      for (int it = 0; it < NV; ++it) {
        const int idx = tid + 512 * it;
        const int row = idx / VPR, v = idx - row * VPR;
        if (n0 + v * 8 >= a.Ks) continue;
        const int r = row / TW, c = row - r * TW;
        const long long off = blk_base + (long long)r * a.out_sh + (long long)c * a.out_sw + n0 + v * 8;
        u32x4 val = *reinterpret_cast<const u32x4*>(smem + row * PITCH + v * 16);
        if (a.addend != nullptr) val = add_bf16x8(val, addv[it]);
        if (a.mask != nullptr) val = lrelu_mask8(val, mskv[it], a.mask_slope);
        *reinterpret_cast<u32x4*>(a.Y + off) = val;
      }
      continue;
    }
    for (int idx = tid; idx < TH * TW * VPR; idx += 512) {
      const int row = idx / VPR, v = idx - row * VPR;
      if (n0 + v * 8 >= a.Ks) continue;
      const int r = row / TW, c = row - r * TW;
      const long long off = blk_base + (long long)r * a.out_sh + (long long)c * a.out_sw;
      *reinterpret_cast<u32x4*>(a.Y + off + n0 + v * 8) = *reinterpret_cast<const u32x4*>(smem + row * PITCH + v * 16);
    }
  }
}

// Two tap programs in one launch (block-uniform branch): <T0A, T1A> for blocks [0, nblk0), <T0B, T1B> for the rest
// (T0B == 0: one program only).
template <int TN, int T0A, int T1A, int T0B, int T1B, int WROWS, int PH, int PW, bool VIRT = false>
__global__ __launch_bounds__(512) void gemm_taps_kernel(const TapsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (T0B > 0) {
    // Two programs read the same patches.  Blocks b and b + 8 share an XCD (speed only: MI355X_MICROARCH.md "XCD placement"),
    // so deal the programs in groups of eight: blocks 16 g .. 16 g + 7 run program 0 on tiles 8 g .. 8 g + 7, blocks
    // 16 g + 8 .. 16 g + 15 program 1 on the same tiles -- the second reader of a patch finds it in its XCD's L2.
    // (nblk0 % 8 == 0 is required by the launcher for this order; otherwise program 1 follows program 0.)
    const int b = (int)blockIdx.x;
    int prog, local;
    if ((a.nblk0 & 7) == 0) {
      prog = (b >> 3) & 1;
      local = ((b >> 4) << 3) | (b & 7);
    } else {
      prog = b >= a.nblk0 ? 1 : 0;
      local = prog ? b - a.nblk0 : b;
    }
    if (prog) {
      gemm_taps_body<TN, T0B, T1B, WROWS, PH, PW>(a, a.prog[1], local, smem);
      return;
    }
    gemm_taps_body<TN, T0A, T1A, WROWS, PH, PW>(a, a.prog[0], local, smem);
  } else {
    gemm_taps_body<TN, T0A, T1A, WROWS, PH, PW, VIRT>(a, a.prog[0], (int)blockIdx.x, smem);
  }
}

}  // namespace jpdse
