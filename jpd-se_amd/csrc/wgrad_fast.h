// Fast bf16 weight-gradient kernel for CDNA4:  dW[k][tap][c] = sum_p dy[p][k] * x[p (+) tap][c].
//
// Both MFMA operands need 8 consecutive PIXELS per lane while memory is channel-contiguous.  The
// generic kernel transposes while staging (8 ds_write_b32 per 32 bytes, VALU packing); here the
// tiles are staged in their natural [pixel][channel] order by LDS-DMA (global_load_lds, 1 KiB per
// wave instruction, source-side swizzle) and transposed for free by ds_read_b64_tr_b16: per group
// of 16 lanes it returns a 4-pixel x 16-channel block column-major (semantics verified on
// hardware by scripts/probe_tr16.hip).  Padding (reflect / zero) is resolved per pixel in the
// loader, so no padded copy of x is made.  One tile = 64*TM k-channels x 64*TN c-channels of one
// filter tap; reduction over pixels in 64-pixel chunks, 2 LDS stages, one barrier per chunk.
//
// LDS image of an operand stage: row = pixel, ROWB = 128*T bytes; the 16-byte slot index is XORed
// with f(pixel) so that the 4 rows touched by one transposed read fall on distinct bank groups:
//   ROWB = 256:  slot ^= (pixel & 3) << 2        ROWB = 128:  slot ^= ((pixel >> 1) & 1) << 2
#pragma once
#include "common.h"
#include "gemm_fast.h"

namespace jpdse {

typedef __attribute__((ext_vector_type(4))) short s16x4;

struct FastWgArgs {
  const bf16_t* X;    // [N][IH][IW][Cs] unpadded
  const bf16_t* DY;   // [M][Ks]
  float* DW;          // fp32 KRSC master-layout gradient
  int M, OH, OW;
  int IH, IW, Cs, C;
  int Ks, K;
  int R, S, sy, sx, py, px, reflect;
  // pixel range cut into `splits` equal parts (block = (tile, split)); a split stores its partial tile into slab
  // `split` of `partial` (DW's layout, slab_stride elements apart) and slab_reduce_kernel adds the slabs in a fixed
  // order -- round 1's stream-K partition met in fp32 atomics (run-to-run nondeterministic)
  int chunks_total, splits, chunks_per_split;
  float* partial;
  long long slab_stride;
  // run mode (input channels not a multiple of 64, e.g. the 40-channel network inputs): X is the
  // materially padded input, a tile's columns are 64*TN consecutive elements of the S*Cs run under
  // filter row r (consecutive taps s are consecutive pixels), taps iterate over r only.
  int run_mode, run_len;
};

template <int ROWB> __device__ __forceinline__ int trswz(int pix) {
  return ROWB >= 256 ? ((pix & 3) << 2) : (ROWB == 128 ? (((pix >> 1) & 1) << 2) : 0);   // 64-B rows: 4 rows = 256 contiguous bytes
}

__device__ __forceinline__ s16x8 tr_frag(const char* p, int row4_bytes) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + row4_bytes));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// The same read as inline asm, for loops that keep LDS-DMA in flight: hipcc treats the tr-read BUILTIN as an
// LDS load that may alias the pending global_load_lds and puts `s_waitcnt vmcnt(0)` in front of the first
// one after every DMA issue (seen in the .s of every kernel using tr_frag: the DMA of the next chunk was
// drained before the current chunk's MFMAs started).  The asm form is invisible to that analysis; the
// caller orders it by hand: counted vmcnt + raw barrier before, tr_wait() (lgkmcnt(0) tied to the
// fragment registers) between the reads and the MFMAs that consume them.
template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ s16x8 tr_frag_asm(uint32_t addr) {
  s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi)
               : "v"(addr), "n"(OFF_LO), "n"(OFF_HI));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// two runtime addresses (row pitch not a compile-time constant)
__device__ __forceinline__ s16x8 tr_frag_asm2(uint32_t addr_lo, uint32_t addr_hi) {
  s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2\n\tds_read_b64_tr_b16 %1, %3"
               : "=&v"(lo), "=&v"(hi)
               : "v"(addr_lo), "v"(addr_hi));
  return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <int N> __device__ __forceinline__ void tr_wait(s16x8 (&f)[N]) {
  if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]));
  else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]));
  else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]));
  else if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
  else {      // longer lists: wait once, tie the rest to the same point
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
#pragma unroll
    for (int i = 4; i < N; ++i) asm volatile("" : "+v"(f[i]));
  }
}

// one 16-pixel k-step of a wave tile: TM + TN transposed fragments (asm reads), then TM*TN MFMAs
template <int TM, int TN, int A_ROWB, int B_ROWB, int KS>
__device__ __forceinline__ void wg_mma_step(uint32_t sbase, const int (&a_tr)[TM], const int (&b_tr)[TN],
                                            f32x16 (&acc)[TM][TN]) {
  s16x8 af[TM], bf[TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) af[i] = tr_frag_asm<KS * 16 * A_ROWB, KS * 16 * A_ROWB + 4 * A_ROWB>(sbase + a_tr[i]);
#pragma unroll
  for (int j = 0; j < TN; ++j) bf[j] = tr_frag_asm<KS * 16 * B_ROWB, KS * 16 * B_ROWB + 4 * B_ROWB>(sbase + b_tr[j]);
  tr_wait(af);
  tr_wait(bf);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
}

// Tile = (WM*TM*32) k-channels x (WN*TN*32) c-channels, WM x WN waves, wave tile 32*TM x 32*TN.
//   <2,2,TM,TN>  64..128 square-ish tiles, 4 waves, 2 blocks per CU (small layers)
//   <2,4,4,2>    256 x 256, 8 waves, 1 block per CU: 2x the FLOPs per staged byte -- the kernel is bound
//                by the L2->LDS fill (ablation: no DMA => 2.2x).
// ABL: timing-only ablation bits (wrong results when non-zero): 1 = no DMA after the first chunk,
// 2 = no barrier, 4 = fragments read once per chunk, 8 = no vmcnt waits
template <int WM, int WN, int TM, int TN, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_fast_kernel(const FastWgArgs a) {
  constexpr int NW = WM * WN;
  constexpr int BKP = 64;                         // pixels per chunk
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int A_ROWB = BM * 2, B_ROWB = BN * 2; // 128, 256 or 512 bytes per pixel row
  constexpr int A_STAGE = BKP * A_ROWB, B_STAGE = BKP * B_ROWB;
  constexpr int STAGE = A_STAGE + B_STAGE;
  constexpr int A_PPU = 1024 / A_ROWB, B_PPU = 1024 / B_ROWB;   // pixels per 1 KiB DMA unit
  constexpr int A_UNITS = BKP / A_PPU, B_UNITS = BKP / B_PPU;
  constexpr int AU = A_UNITS / NW, BU = B_UNITS / NW;           // per wave
  static_assert(A_UNITS % NW == 0 && B_UNITS % NW == 0, "DMA units must divide evenly over the waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const uint32_t lds0 = lds_addr_of(smem);

  const int c_tiles = ((a.run_mode ? a.run_len : a.Cs) + BN - 1) / BN;
  const int n_taps = a.run_mode ? a.R : a.R * a.S;
  // block -> (k tile, tap, c tile, split).  The taps of one (k tile, c tile, split) stage the same dy pixels and nearly
  // the same input pixels: they get consecutive slots of ONE XCD (blocks b, b + 8, ... share an XCD: observed dispatch,
  // speed only), so that its L2 serves all of them -- in plain tile order they ran apart in time and every tap re-fetched
  // both operands from beyond L2 (20x the algorithmic bytes on the 16-tap PatchGAN layers).
  const int k_tiles = (a.Ks + BM - 1) / BM;
  const int groups = k_tiles * c_tiles * a.splits;
  int grp, tap;
  if ((groups & 7) == 0) {
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    tap = j % n_taps;
    grp = (j / n_taps) * 8 + xcd;
  } else {
    tap = blockIdx.x % n_taps;
    grp = blockIdx.x / n_taps;
  }
  const int split = grp % a.splits, kc = grp / a.splits;
  const int ct = kc % c_tiles, kt = kc / c_tiles;
  const int ch_begin = split * a.chunks_per_split;
  int ch_end = ch_begin + a.chunks_per_split;
  ch_end = ch_end < a.chunks_total ? ch_end : a.chunks_total;
  float* const out = a.splits > 1 ? a.partial + (long long)split * a.slab_stride : a.DW;
  const int r = a.run_mode ? tap : tap / a.S, s = a.run_mode ? 0 : tap - r * a.S;
  const int k0 = kt * BM, c0 = ct * BN;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);

  // ---- DMA unit state -----------------------------------------------------------------------
  // A' (dy): unit u covers pixels [u*A_PPU, +A_PPU) of the chunk; lane -> (pixel in unit, 16-B slot)
  int a_pix[AU], a_soff[AU], a_lds[AU];
#pragma unroll
  for (int i = 0; i < AU; ++i) {
    const int u = wid + NW * i;
    const int pl = lane / (A_ROWB / 16), slot = lane % (A_ROWB / 16);
    const int pix = u * A_PPU + pl;
    a_pix[i] = pix;
    int ch = k0 + ((slot ^ trswz<A_ROWB>(pix)) << 3);
    ch = ch + 8 <= a.Ks ? ch : a.Ks - 8;                 // tile overhang: rows >= K are masked at the store
    a_soff[i] = ch;
    a_lds[i] = u * 1024;
  }
  // B' (x): same, plus the tap-shifted, padded source pixel
  int b_pix[BU], b_soff[BU], b_lds[BU];
  int bn[BU], boh[BU], bow[BU];                          // cursor of the unit's pixel: (n, oh, ow)
#pragma unroll
  for (int i = 0; i < BU; ++i) {
    const int u = wid + NW * i;
    const int pl = lane / (B_ROWB / 16), slot = lane % (B_ROWB / 16);
    const int pix = u * B_PPU + pl;
    b_pix[i] = pix;
    int ch = c0 + ((slot ^ trswz<B_ROWB>(pix)) << 3);
    if (!a.run_mode) ch = ch + 8 <= a.Cs ? ch : a.Cs - 8;   // run mode: overhang reads the zeroed slack
    b_soff[i] = ch;
    b_lds[i] = A_STAGE + u * 1024;
    int p = ch_begin * BKP + pix;
    p = p < a.M ? p : a.M - 1;
    bow[i] = p % a.OW;
    const int t = p / a.OW;
    boh[i] = t % a.OH;
    bn[i] = t / a.OH;
  }

  // ---- transposed fragment read offsets -----------------------------------------------------
  // lane: g = lane>>4 -> (h = g>>1: which 8 pixels of the 16-pixel k-step, cb = g&1: 16-channel block),
  //       li = lane&15 -> q = li>>2 (row of the 4x16 block), p = li&3 (4-channel column group)
  int a_tr[TM], b_tr[TN];
  {
    const int g = lane >> 4, li = lane & 15, h = g >> 1, cb = g & 1, q = li >> 2, p = li & 3;
    const int pix = 8 * h + q;                          // + 16*ks + 4*half are multiples of 4: swizzle unchanged
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ch = (wm * TM + i) * 32 + cb * 16 + 4 * p;          // channel within the tile
      a_tr[i] = pix * A_ROWB + ((((ch >> 3) ^ trswz<A_ROWB>(pix)) << 4) | ((ch & 7) << 1));
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int ch = (wn * TN + j) * 32 + cb * 16 + 4 * p;
      b_tr[j] = A_STAGE + pix * B_ROWB + ((((ch >> 3) ^ trswz<B_ROWB>(pix)) << 4) | ((ch & 7) << 1));
    }
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // Loader address arithmetic is incremental and 32-bit (element offsets < 2^31, checked on the host):
  // with 64-bit multiplies per unit it cost 150-230 VALU slots per chunk against 8-32 MFMAs.
  int a_off[AU], x_off[BU];
#pragma unroll
  for (int i = 0; i < AU; ++i) a_off[i] = (ch_begin * BKP + a_pix[i]) * a.Ks + a_soff[i];
#pragma unroll
  for (int i = 0; i < BU; ++i)
    x_off[i] = ((bn[i] * a.IH + boh[i] * a.sy) * a.IW + bow[i] * a.sx) * a.Cs + b_soff[i];
  const int tap_dh = r - a.py, tap_dw = s - a.px;
  const int tap_off = (tap_dh * a.IW + tap_dw) * a.Cs;
  const int row_el = a.IW * a.Cs;                        // elements per input row
  const int adv_w = BKP * a.sx * a.Cs, wrap_w = a.OW * a.sx * a.Cs, adv_h = a.sy * row_el;
  const int wrap_h = a.OH * a.sy * row_el - a.IH * row_el;   // leaving the last output row of an image
  int ich = ch_begin;       // next chunk to issue
  const int IHm1 = a.IH - 1, IWm1 = a.IW - 1;
  const bool use_refl = !a.run_mode && a.reflect;          // wave-uniform
  const bool no_test = a.run_mode || a.reflect;            // every pixel readable: no zero page
  const bool narrow = a.OW < BKP;
  auto issue = [&](int stage) {
    char* const st = smem + stage * STAGE;
    const int pbase = ich * BKP;
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      const bf16_t* src = pbase + a_pix[i] < a.M ? a.DY + a_off[i] : zero;
      glds16(src, st + a_lds[i]);
      a_off[i] += BKP * a.Ks;
    }
    // Branch-free (the if / else-if padding chain and the per-lane `while` of the cursor compiled to exec-mask loops
    // behind every DMA): reflected coordinates are the identity for in-range pixels, so they are computed for every
    // mode and only applied / tested through the wave-uniform flags; with OW >= 64 the cursor wraps at most once.
#pragma unroll
    for (int i = 0; i < BU; ++i) {
      const int ih = boh[i] * a.sy + tap_dh, iw = bow[i] * a.sx + tap_dw;
      int ih2 = ih < 0 ? -ih : ih, iw2 = iw < 0 ? -iw : iw;
      ih2 = ih2 > IHm1 ? 2 * IHm1 - ih2 : ih2;
      iw2 = iw2 > IWm1 ? 2 * IWm1 - iw2 : iw2;
      const int fix = __mul24(ih2 - ih, row_el) + __mul24(iw2 - iw, a.Cs);
      const int off = x_off[i] + tap_off + (use_refl ? fix : 0);
      const bool ok = no_test | (((unsigned)ih < (unsigned)a.IH) & ((unsigned)iw < (unsigned)a.IW));
      const bf16_t* const src = a.X + (ok ? off : 0);
      glds16(ok ? src : zero, st + b_lds[i]);
      // advance the cursor by one chunk (clamped at the last pixel: its dy row is the zero page)
      const bool adv = pbase + BKP + b_pix[i] < a.M;
      int nbow = bow[i] + BKP, nboh = boh[i], nx = x_off[i] + adv_w;
      {
        const bool w1 = nbow >= a.OW;
        nbow -= w1 ? a.OW : 0;
        nx += w1 ? adv_h - wrap_w : 0;
        nboh += w1 ? 1 : 0;
        const bool h1 = nboh == a.OH;
        nboh = h1 ? 0 : nboh;
        nx -= h1 ? wrap_h : 0;
      }
      if (narrow) {                                   // wave-uniform: output rows shorter than a chunk
        while (nbow >= a.OW) {
          nbow -= a.OW;
          nx += adv_h - wrap_w;
          if (++nboh == a.OH) { nboh = 0; nx -= wrap_h; }
        }
      }
      bow[i] = adv ? nbow : bow[i];
      boh[i] = adv ? nboh : boh[i];
      x_off[i] = adv ? nx : x_off[i];
    }
    ++ich;
  };

  int stage = 0;
  if (ch_begin < ch_end) issue(0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  for (int c = ch_begin; c < ch_end; ++c) {
    if (c + 1 < ch_end && !(ABL & 1)) issue(stage ^ 1);
    const uint32_t sbase = lds0 + stage * STAGE;
    __builtin_amdgcn_s_setprio(1);
    wg_mma_step<TM, TN, A_ROWB, B_ROWB, 0>(sbase, a_tr, b_tr, acc);
    wg_mma_step<TM, TN, A_ROWB, B_ROWB, (ABL & 4) ? 0 : 1>(sbase, a_tr, b_tr, acc);
    wg_mma_step<TM, TN, A_ROWB, B_ROWB, (ABL & 4) ? 0 : 2>(sbase, a_tr, b_tr, acc);
    wg_mma_step<TM, TN, A_ROWB, B_ROWB, (ABL & 4) ? 0 : 3>(sbase, a_tr, b_tr, acc);
    __builtin_amdgcn_s_setprio(0);
    if (!(ABL & 8)) wait_vmcnt<0>();
    if (!(ABL & 2)) __builtin_amdgcn_s_barrier();
    stage ^= 1;
  }

  // ---- epilogue: scatter into the KRSC master-layout gradient -------------------------------
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    int cc = c0 + (wn * TN + j) * 32 + (lane & 31);
    int s_out = s;
    if (a.run_mode) {               // column of the run -> (tap s, channel c)
      if (cc >= a.run_len) continue;
      s_out = cc / a.Cs;
      cc -= s_out * a.Cs;
    }
    if (cc >= a.C) continue;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int k = k0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        if (k >= a.K) continue;
        out[(((long long)k * a.R + r) * a.S + s_out) * a.C + cc] = acc[i][j][e];
      }
    }
  }
}

// dz[p'][k*R*S + r*S + s] = dy[p' - (r,s)][k]  (zero outside): with it the weight gradient of a conv with
// very few output channels (the 64->3 head, the 512->1 PatchGAN map) becomes a dense 1x1 weight
// gradient  dW[(k,r,s)][c] = sum_p' dz[p'][(k,r,s)] * xpad[p'][c]  whose row order is already the KRSC
// master layout -- no 32-row MFMA tile wasted on 3 live rows.  Stride 1 only.
__global__ void expand_dy_taps_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dz, int OH, int OW, int Ks,
                                      int K, int R, int S, int Hp, int Wp, int Kexp_s, long long total_vec) {
  const int cv = Kexp_s / 8;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total_vec;
       idx += (long long)gridDim.x * blockDim.x) {
    const int v = (int)(idx % cv);
    long long t = idx / cv;
    const int wp = (int)(t % Wp);
    t /= Wp;
    const int hp = (int)(t % Hp);
    const int n = (int)(t / Hp);
    uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ch = v * 8 + e;
      const int k = ch / (R * S), tap = ch - k * (R * S);
      const int r = tap / S, s = tap - r * S;
      const int oh = hp - r, ow = wp - s;
      uint32_t val = 0;
      if (k < K && (unsigned)oh < (unsigned)OH && (unsigned)ow < (unsigned)OW)
        val = dy[(((long long)n * OH + oh) * OW + ow) * Ks + k];
      w[e >> 1] |= val << ((e & 1) * 16);
    }
    u32x4 o = {w[0], w[1], w[2], w[3]};
    *reinterpret_cast<u32x4*>(dz + idx * 8) = o;
  }
}

}  // namespace jpdse
