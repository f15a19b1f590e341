// 3x3 convolutions over 64-channel inputs at full resolution (stride 1 and 2; forward, and the data gradient of the
// zero-padded stride-1 ones): G's first down-sampling conv 64 -> 128 at 512x1024, VGG19 conv1_2 / conv2_1 and their data
// gradients.  K-dim = 9 * 64 = 576: nine K-tiles.  On the tiled GEMM kernels these layers are bound by the L2 -> LDS
// fill, not by MFMA or HBM: every 128x128 tile re-stages the 147 KB filter and each input pixel once per tap
// (1.2 GB of fill for 0.4 GB of HBM traffic, ~24 GB/s per CU with a 9-step loop that never fills its pipeline).
//
// Here the FILTER LIVES IN REGISTERS and the input streams through LDS exactly once:
//   * a block owns a 64-pixel-wide strip of TH output rows of one image and walks down it, one output row per iteration;
//   * wave (wc, wp) owns 32 output channels (wc) of 64 / WP pixels (wp) and keeps its 32 x 576 filter slice as 36
//     MFMA B-fragments (144 VGPRs), loaded once per block straight from the packed panel;
//   * input rows arrive by LDS-DMA into a ring of NR rows, issued LA iterations ahead (counted vmcnt, one raw barrier per
//     output row); padding is resolved by the loader (zero page); for stride 2 the loader de-interleaves a row into
//     [even pixels | odd pixels] so that a tap's 32 pixels are 32 CONSECUTIVE 128-byte LDS rows and the halo kernel's
//     conflict-free swizzle (16-byte chunk ^ (row >> 1)) applies unchanged;
//   * per output row and wave: 9 taps x 4 k-steps x MT MFMAs, A fragments read at tap-shifted addresses; the result goes
//     through a per-wave LDS tile (no block barrier) to 16-byte channel-vector stores, 64 contiguous bytes per pixel and wave.
// LDS fill per output row: STRIDE new input rows (8.4 / 33 KB) instead of 9 x 32 KB per 128 pixels.
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "gemm_halo.h"
#include "head_fwd.h"

namespace jpdse {

__device__ __forceinline__ void lds_store32u(uint32_t addr, uint32_t v) {
  asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

// acc_tile_to_lds (gemm_fast.h) with the LDS stores as inline asm: an ordinary LDS store issued while an LDS-DMA is in
// flight gets `s_waitcnt vmcnt(0)` from hipcc (it cannot prove that the two do not overlap), which would drain the row
// prefetch at every output row.  The values pass through VALU (bias, activation, bf16 pack) on their way from the MFMA
// accumulators, so the compiler's own MFMA -> VALU wait states apply (see head_fwd.h).
// bv: the lane's bias (0 for dead columns), nslope: factor for negative values (1 = no activation, 0 = ReLU, slope =
// LeakyReLU), keep: 0 for columns beyond Kout -- all loop-invariant, computed once per block.
template <int TM>
__device__ __forceinline__ void acc_rows_to_lds(uint32_t tile, int pitch, int lane, const f32x16 (&acc)[TM][1], float bv,
                                                float nslope, float keep) {
  const int odd = lane & 1;
  const int lcol = lane & 31;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int ep = 0; ep < 8; ++ep) {
      const int e = 2 * ep;
      float v0 = acc[i][0][e] + bv, v1 = acc[i][0][e + 1] + bv;
      v0 = (v0 > 0.f ? v0 : v0 * nslope) * keep;
      v1 = (v1 > 0.f ? v1 : v1 * nslope) * keep;
      // neighbour exchange as a DPP quad permutation [1,0,3,2]: __shfl_xor is a ds_bpermute round trip, 16 of them per
      // output row with nothing to hide them behind (one wave per SIMD)
      const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, odd ? v0 : v1), 0xB1, 0xF, 0xF, false));
      const float lo = odd ? recv : v0, hi = odd ? v1 : recv;
      const uint32_t word = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
      const int row = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) + odd;
      lds_store32u(tile + row * pitch + (lcol - odd) * 2, word);
    }
  }
}

struct RowsArgs {
  const bf16_t* X;     // [N][IH][IW][64]
  const bf16_t* B;     // panel [b_rows][9 * 64], row = output channel, (r, s, c) contiguous
  const float* bias;
  bf16_t* Y;
  int N, OH, OW, IH, IW;
  int py, px;          // ih = oh * STRIDE + r - py
  int Kout, Ks, b_rows;
  long long out_sn, out_sh, out_sw, out_base;
  int act;
  float slope;
  const bf16_t* mask;    // optional fused ReLU backward of the conv's INPUT (data gradient), Y's addressing
  const bf16_t* addend;  // optional: Y = result + addend, before the mask
  int TH, bands, strips;
  int n_tiles;           // Ks / (32 * WC) column tiles (blockIdx fastest but one)
  float* mom;            // optional: per-block (mean, M2) of y per channel (common.h) for the InstanceNorm that follows,
  int mom_slots;         //   [N][Ks][mom_slots][2], slot = (band * strips + strip) * WP + wp (bias NULL, no activation)
};

template <int STRIDE, int WC> struct RowsGeom {
  static constexpr int MT = WC / 2;                           // 32-row m-tiles per wave (WC = 4: one pixel group of 64; WC = 2: two of 32)
  static constexpr int PIX = STRIDE == 1 ? 66 : 129;          // staged pixels per input row
  static constexpr int UNITS = (PIX + 7) / 8;                 // 1 KiB DMA units per row
  static constexpr int ROWB = UNITS * 1024;
  // Ring rows / iterations of look-ahead.  What bounds these kernels once the fragment reads are pipelined is the number of
  // bytes in flight per CU (HBM latency x bandwidth): the ring is as deep as LDS allows -- two blocks per CU for the
  // 64-output stride-1 layers (75 KB each), one block with the whole LDS otherwise.
  static constexpr int NR = STRIDE == 2 ? 8 : (WC == 2 ? 7 : 12);
  static constexpr int LA = STRIDE == 2 ? 2 : (WC == 2 ? 3 : 8);
  static_assert(STRIDE * LA + 3 <= NR, "rows in use + rows in flight fit the ring");
  static constexpr int ODD0 = 65;                             // stride 2: LDS pixel index of the first odd padded column
  static constexpr int WPITCH = 64;                           // per-wave epilogue tile: pixels x 64 bytes, unpadded: the packed dword
                                                              // stores (even lanes row r, odd lanes row r+1) and the 16-byte reads (4 rows x
                                                              // 4 quarters per lane group) are both conflict free; pitch 80 cost 16 % of the LDS cycles
  static constexpr int WTILE = MT * 32 * WPITCH;
  static constexpr int LDS = NR * ROWB + 4 * WTILE;
};

// s_waitcnt vmcnt(BASE + k * STEP), k = 0 .. MAXK chosen at run time (the immediate must be a constant)
template <int BASE, int STEP, int MAXK = 8> __device__ __forceinline__ void wait_vmcnt_sel(int k) {
  static_assert(MAXK <= 8 && BASE + MAXK * STEP <= 63, "vmcnt is a 6-bit counter");
  constexpr auto at = [](int j) constexpr { return BASE + (j < MAXK ? j : MAXK) * STEP; };
  switch (k) {
    case 0: wait_vmcnt<at(0)>(); break;
    case 1: wait_vmcnt<at(1)>(); break;
    case 2: wait_vmcnt<at(2)>(); break;
    case 3: wait_vmcnt<at(3)>(); break;
    case 4: wait_vmcnt<at(4)>(); break;
    case 5: wait_vmcnt<at(5)>(); break;
    case 6: wait_vmcnt<at(6)>(); break;
    case 7: wait_vmcnt<at(7)>(); break;
    default: wait_vmcnt<at(8)>(); break;
  }
}

template <int STRIDE, int WC, bool FUSED>
__global__ __launch_bounds__(256, ((STRIDE == 1 && WC == 2) ? 2 : 1)) void conv_rows_kernel(const RowsArgs a) {
  typedef RowsGeom<STRIDE, WC> G;
  constexpr int WP = 4 / WC, MW = 64 / WP, MT = G::MT;        // pixel groups, pixels per wave, 32-row m-tiles per wave
  static_assert(MT == MW / 32 && G::LA <= 8, "geometry");
  constexpr int U0 = G::UNITS / 4, U1 = U0 + 1, EXTRA = G::UNITS % 4;
  constexpr int NST = MT * 2;                                 // 16-byte stores per lane and output row
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wid % WC, wp = wid / WC;
  int b = blockIdx.x;
  const int tile_n = b % a.n_tiles; b /= a.n_tiles;
  const int strip = b % a.strips; b /= a.strips;
  const int band = b % a.bands;
  const int n = b / a.bands;
  const int oh0 = band * a.TH, ow0 = strip * 64;
  const int ncol0 = tile_n * 32 * WC + wc * 32;               // this wave's first output channel
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  char* const wtile = smem + G::NR * G::ROWB + wid * G::WTILE;
  const uint32_t smem0 = lds_addr32(smem);

  // ---- loader state: this wave's DMA units of a row (u = wid, wid + 4, ...); per lane the source column offset
  int col_off[U1];
#pragma unroll
  for (int k = 0; k < U1; ++k) {
    const int u = wid + 4 * k;
    const int lp = u * 8 + (lane >> 3);
    int pp;                                                   // padded column of LDS pixel lp
    if (STRIDE == 1) pp = lp;
    else pp = lp < G::ODD0 ? 2 * lp : 2 * (lp - G::ODD0) + 1;
    const int iw = ow0 * STRIDE - a.px + pp;
    const bool ok = u < G::UNITS && lp < G::PIX && (unsigned)iw < (unsigned)a.IW;
    const int chunk = ((lane & 7) ^ (lp >> 1)) & 7;
    col_off[k] = ok ? iw * 64 + chunk * 8 : -1;
  }
  const bf16_t* const ximg = a.X + (long long)n * a.IH * a.IW * 64;
  const int ih_base = oh0 * STRIDE - a.py;
  const int row_elems = a.IW * 64;
  auto issue_row = [&](int jr, int slot) {                    // input row jr of the band -> ring slot
    const int ih = ih_base + jr;
    const bool row_ok = (unsigned)ih < (unsigned)a.IH;
    const bf16_t* const xrow = ximg + (row_ok ? ih : 0) * (long long)row_elems;
    char* const dst = smem + slot * G::ROWB;
#pragma unroll
    for (int k = 0; k < U1; ++k) {
      if (k < U0 || wid < EXTRA) {
        const bf16_t* src = (row_ok && col_off[k] >= 0) ? xrow + col_off[k] : zero;
        glds16(src, dst + (wid + 4 * k) * 1024);
      }
    }
  };

  // ---- prologue: rows of iterations 0 .. LA-1, then the filter slice into registers (under the DMA latency)
  constexpr int PRO = STRIDE * (G::LA - 1) + 3;
#pragma unroll
  for (int jr = 0; jr < PRO; ++jr) issue_row(jr, jr);
  s16x8 breg[36];
  {
    int brow = ncol0 + (lane & 31);
    brow = brow < a.b_rows ? brow : a.b_rows - 1;
    const bf16_t* const bp = a.B + (long long)brow * 576 + (lane >> 5) * 8;
#pragma unroll
    for (int t = 0; t < 36; ++t) breg[t] = *reinterpret_cast<const s16x8*>(bp + t * 16);
  }
  // A-fragment addressing: LDS pixel of this lane's row for tap column s, per m-tile
  int a_base[3][MT], a_sw[3][MT];
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int q = wp * MW + mt * 32 + (lane & 31);
      const int lp = STRIDE == 1 ? q + s : ((s & 1) ? G::ODD0 : 0) + q + (s >> 1);
      a_base[s][mt] = lp << 7;
      a_sw[s][mt] = ((lp >> 1) & 7) << 4;
    }
  const int hsel = lane >> 5;
  const bool live_col = ncol0 + (lane & 31) < a.Kout;
  const float bv = (a.bias != nullptr && live_col) ? a.bias[ncol0 + (lane & 31)] : 0.f;
  const float nslope = a.act == JPDSE_ACT_RELU ? 0.f : (a.act == JPDSE_ACT_LRELU ? a.slope : 1.f);   // no Tanh here (launcher)
  const float keep = live_col ? 1.f : 0.f;
  // keep the filter fragments out of the loop's reach of rematerialisation: they are loaded exactly once
#pragma unroll
  for (int t = 0; t < 36; ++t) asm volatile("" : "+v"(breg[t]));

  float ms1 = 0.f, ms2 = 0.f, mpilot = 0.f;                   // moments of this lane's column over the block's rows, about a pilot (common.h)
  int base = 0;                                               // ring slot of input row STRIDE * i
  int nslot = PRO % G::NR;                                    // ring slot of the next row to issue
  int njr = PRO;
  for (int i = 0; i < a.TH; ++i) {
    // rows of iteration i landed; in flight behind them: the rows of iterations i+1 .. i+LA-1 and the stores of up to LA
    // earlier iterations (fewer at the start) -- see the issue order below
    {
      constexpr int B1 = (G::LA - 1) * STRIDE * U1, B0 = (G::LA - 1) * STRIDE * U0;
      if constexpr (FUSED) {
        // i > 0: the epilogue of iteration i-1 waited for its own addend / mask loads, which are younger than these rows
        if (i == 0) { if (wid < EXTRA) wait_vmcnt<B1>(); else wait_vmcnt<B0>(); }
      } else {
        const int k = i < G::LA ? i : G::LA;                  // iterations whose NST stores may still be in flight
        if (wid < EXTRA) wait_vmcnt_sel<B1, NST, G::LA>(k); else wait_vmcnt_sel<B0, NST, G::LA>(k);
      }
    }
    __builtin_amdgcn_s_barrier();     // every wave's pieces of these rows landed; the slots issued below were last read in i-1
    asm volatile("" ::: "memory");
    const int oh = oh0 + i;
    const long long orow = a.out_base + n * a.out_sn + (long long)oh * a.out_sh + (long long)(ow0 + wp * MW) * a.out_sw + ncol0;
    u32x4 addv[NST], mskv[NST];
    if constexpr (FUSED) {
#pragma unroll
      for (int t = 0; t < NST; ++t) {
        const int v = lane + 64 * t;
        const long long off = orow + (long long)(v >> 2) * a.out_sw + (v & 3) * 8;
        if (a.addend != nullptr) addv[t] = *reinterpret_cast<const u32x4*>(a.addend + off);
        if (a.mask != nullptr) mskv[t] = *reinterpret_cast<const u32x4*>(a.mask + off);
      }
    }
#pragma unroll
    for (int k = 0; k < STRIDE; ++k) {                        // rows of iteration i + LA
      issue_row(njr, nslot);
      ++njr;
      nslot = nslot + 1 == G::NR ? 0 : nslot + 1;
    }
    f32x16 acc[MT][1];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mt][0][e] = 0.f;
    // One wave per SIMD: nothing else hides the LDS round trip, and hipcc's own schedule (read, wait lgkmcnt(0), MFMA)
    // exposed it at every one of the 36 k-steps (0.16 ms per launch at 512x1024, mostly waiting).  The fragment reads are
    // inline asm, issued DEPTH k-steps ahead and waited for with a counted lgkmcnt (LDS operations return in order; the
    // loop has no scalar-memory loads sharing the counter).
    uint32_t rbase[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      int slot = base + r;
      slot = slot >= G::NR ? slot - G::NR : slot;
      rbase[r] = smem0 + slot * G::ROWB;
    }
    constexpr int DEPTH = 3;                                  // k-steps of fragments in flight
    s16x8 fr[DEPTH + 1][MT];
    auto rd = [&](int t, s16x8 (&f)[MT]) {                    // t = tap * 4 + ks, compile-time after unrolling
      const int tap = t >> 2, ks = t & 3, r = tap / 3, sx = tap - 3 * r;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        f[mt] = lds_read128_asm(rbase[r] + a_base[sx][mt] + (((2 * ks + hsel) << 4) ^ a_sw[sx][mt]));
    };
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) rd(t, fr[t]);
#pragma unroll
    for (int t = 0; t < 36; ++t) {
      if (t + DEPTH < 36) rd(t + DEPTH, fr[(t + DEPTH) % (DEPTH + 1)]);
      s16x8 (&f)[MT] = fr[t % (DEPTH + 1)];
      const int behind = (36 - 1 - t) < DEPTH ? (36 - 1 - t) : DEPTH;      // k-steps issued after step t
      if constexpr (MT == 2) {
        if (behind == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f[0]), "+v"(f[1]));
        else if (behind == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0]), "+v"(f[1]));
        else if (behind == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[0]), "+v"(f[1]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]));
      } else {
        if (behind == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f[0]));
        else if (behind == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[0]));
        else if (behind == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f[0]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]));
      }
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[mt], breg[t], acc[mt][0], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(0);
    base += STRIDE;
    base = base >= G::NR ? base - G::NR : base;
    if (a.mom != nullptr) {
      if (i == 0) mpilot = bf16_round(acc[0][0][0]);          // this lane's own pilot (no cross-lane traffic in this loop)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const float v = bf16_round(acc[mt][0][e]) - mpilot;
          ms1 += v;
          ms2 += v * v;
        }
    }
    // ---- epilogue of this output row: per-wave LDS tile (LDS operations of one wave execute in order), 16-byte stores
    acc_rows_to_lds<MT>(lds_addr32(wtile), G::WPITCH, lane, acc, bv, nslope, keep);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (FUSED) {                                    // addend / mask loads are older than this iteration's row DMA
      if (wid < EXTRA) wait_vmcnt<STRIDE * U1>(); else wait_vmcnt<STRIDE * U0>();
    }
    // (the launcher guarantees OH % TH == 0: every iteration issues its NST stores, which the counted waits rely on)
#pragma unroll
    for (int t = 0; t < NST; ++t) {
      const int v = lane + 64 * t;
      u32x4 val = *reinterpret_cast<const u32x4*>(wtile + (v >> 2) * G::WPITCH + (v & 3) * 16);
      const long long off = orow + (long long)(v >> 2) * a.out_sw + (v & 3) * 8;
      if constexpr (FUSED) {
        if (a.addend != nullptr) val = add_bf16x8(val, addv[t]);
        if (a.mask != nullptr) val = relu_mask8(val, mskv[t]);
      }
      *reinterpret_cast<u32x4*>(a.Y + off) = val;
    }
  }
  if (a.mom != nullptr) {
    // lanes l and l + 32 hold the same column (the other half of the pixel rows): each turns its sums into (mean, M2) over
    // its TH * MW / 2 values, then the pair is merged
    const float half = 0.5f * (float)(a.TH * MW);
    float mean, m2;
    shifted_to_mean_m2(ms1, ms2, mpilot, half, mean, m2);
    const float mean_o = __shfl_xor(mean, 32, 64), m2_o = __shfl_xor(m2, 32, 64);
    chan_merge_equal(mean, m2, mean_o, m2_o, half);
    const int col = ncol0 + lane;
    if (lane < 32 && col < a.Ks) {
      const int slot = (band * a.strips + strip) * WP + wp;
      float* const o = a.mom + (((long long)n * a.Ks + col) * a.mom_slots + slot) * 2;
      o[0] = mean;
      o[1] = m2;
    }
  }
}

}  // namespace jpdse
