// Second A/B experiment of round 4 on the halo kernel's wave tile (developer build only), the opposite direction of gemm_halo4.h:
// SIXTEEN waves -- four per SIMD -- with a 64-pixel x 32-channel wave tile (one image row of the 4 x 64 patch x a quarter of the 128
// output channels: 2 accumulator tiles = 32 registers).  The four-wave form showed that fewer LDS reads do not pay when nothing
// covers a wave's own latencies (-15 %); this one spends 50 % MORE fragment reads (2 A + 1 B per 2 MFMAs) to have four waves per SIMD
// take turns at the matrix pipe behind the one barrier per tap.  Same block tile, same staging.  Plain forward only.
// Measured: profiles/r04_halo_wavetile_ab.txt.
#pragma once
#include "gemm_halo.h"

namespace jpdse {

__global__ __launch_bounds__(1024) void gemm_halo16_kernel(const HaloArgs a) {
  constexpr int TH = 4, TN = 1, R = 3, S = 3, TAPS = 9;
  constexpr int NW = 16, WN = 4;
  constexpr int BN = WN * TN * 32;                            // 128
  constexpr int PH = TH + R - 1, PW = 64 + S - 1, NP = PH * PW;
  constexpr int UH = (NP + 7) / 8;
  constexpr int HALO = UH * 1024;
  constexpr int B_STAGE = BN * 128;
  constexpr int B_UNITS = BN / 8, BU = B_UNITS / NW;          // 1 weight piece per wave and tap
  constexpr int HU = (UH + NW - 1) / NW;                      // 4 patch units per wave
  static_assert(TAPS - 2 >= HU, "one patch unit of the next slab per tap over taps 0..HU-1");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const halo0 = smem;
  char* const bring = smem + 2 * HALO;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int tiles_w = a.OW / 64, tiles_h = a.OH / TH;
  const int tiles_m = a.N * tiles_h * tiles_w;
  const int bid = blockIdx.x;
  const int tile_m = bid % tiles_m, tile_n = bid / tiles_m;
  const int tw_i = tile_m % tiles_w, t1 = tile_m / tiles_w;
  const int th_i = t1 % tiles_h, n = t1 / tiles_h;
  const int oh0 = th_i * TH, ow0 = tw_i * 64, n0 = tile_n * BN;
  const int lrow = lane >> 3, lslot = lane & 7;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);

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
    const int swz_p = p;
    p = p < NP ? p : NP - 1;
    const int hr = p / PW, wc = p - hr * PW;
    int ih = oh0 - a.py + hr, iw = ow0 - a.px + wc;
    bool ok = true;
    if (a.reflect) {
      ih = ih < 0 ? -ih : (ih >= a.IH ? 2 * (a.IH - 1) - ih : ih);
      iw = iw < 0 ? -iw : (iw >= a.IW ? 2 * (a.IW - 1) - iw : iw);
    } else {
      ok = ((unsigned)ih < (unsigned)a.IH) & ((unsigned)iw < (unsigned)a.IW);
    }
    const int chunk = (lslot ^ (swz_p >> 1)) & 7;
    h_off[i] = ok ? (((long long)n * a.IH + ih) * a.IW + iw) * a.Cs + chunk * 8 : -1;
    h_lds[i] = ug * 1024;
  }
  const bf16_t* b_ptr[BU];
  int b_lds[BU];
  const long long ktot = (long long)TAPS * a.Cs;
#pragma unroll
  for (int j = 0; j < BU; ++j) {
    const int u = wid + j * NW;
    const int row = u * 8 + lrow;
    int br = n0 + row;
    br = br < a.b_rows ? br : a.b_rows - 1;
    b_ptr[j] = a.B + (long long)br * ktot + ((lslot ^ (row >> 1)) & 7) * 8;
    b_lds[j] = u * 1024;
  }

  constexpr int FM = 2, FN = TN, KS = 4;
  int pb[FM];
#pragma unroll
  for (int i = 0; i < FM; ++i) pb[i] = wm * PW + i * 32 + (lane & 31);
  const int hsel = lane >> 5;
  int b_rd[FN][KS];
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int row = wn * TN * 32 + j * 32 + (lane & 31);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) b_rd[j][ks] = swz128(row, 2 * ks + hsel);
  }
  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  const int CC = a.Cs >> 6;
  auto issue_patch_unit = [&](int i, int slab) {
    char* const dst = halo0 + (slab & 1) * HALO + h_lds[i];
    const bf16_t* src = h_off[i] >= 0 ? a.X + h_off[i] + slab * 64 : zero;
    glds16(src, dst);
  };
  auto issue_b = [&](int tile) {
    const int slab = tile / TAPS, tap = tile - slab * TAPS;
    char* const st = bring + (tile % 3) * B_STAGE;
    const long long koff = (long long)tap * a.Cs + slab * 64;
#pragma unroll
    for (int j = 0; j < BU; ++j) glds16(b_ptr[j] + koff, st + b_lds[j]);
  };
#pragma unroll
  for (int i = 0; i < HU; ++i)
    if (i < n_hu) issue_patch_unit(i, 0);
  issue_b(0);
  issue_b(1);

  auto tap_body = [&](const int tap_r, const int tap_s, const int slab, const bool more, const char* const hb) {
    const int tap = tap_r * S + tap_s;
    if (tap < TAPS - 1 || more) wait_vmcnt<BU>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    // issue group t+2: one patch unit of the NEXT slab, then weight tile t+2
    if (tap < HU && tap < n_hu && more) {
#pragma unroll
      for (int i = 0; i < HU; ++i)
        if (i == tap) issue_patch_unit(i, slab + 1);
    }
    const int tap2 = tap + 2 < TAPS ? tap + 2 : tap + 2 - TAPS;
    if (tap + 2 < TAPS || more) {
      char* const st2 = bring + ((tap_s + 2) % 3) * B_STAGE;
      const long long koff = (long long)tap2 * a.Cs + (slab + (tap + 2 < TAPS ? 0 : 1)) * 64;
#pragma unroll
      for (int jj = 0; jj < BU; ++jj) glds16(b_ptr[jj] + koff, st2 + b_lds[jj]);
    }
    const char* const st = bring + tap_s * B_STAGE;
    int tapoff = tap_r * PW + tap_s;
    asm volatile("" : "+s"(tapoff));       // keeps the per-tap fragment addresses out of hipcc's loop-invariant hoisting (72 VGPRs otherwise: spills)
    int a_base[FM], a_sw[FM];
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int pt = pb[i] + tapoff;
      a_base[i] = pt << 7;
      a_sw[i] = ((pt >> 1) & 7) << 4;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s16x8 af[FM], bf[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = *reinterpret_cast<const s16x8*>(hb + a_base[i] + (((2 * ks + hsel) << 4) ^ a_sw[i]));
#pragma unroll
      for (int j = 0; j < FN; ++j) bf[j] = *reinterpret_cast<const s16x8*>(st + b_rd[j][ks]);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  };
  for (int slab = 0; slab < CC; ++slab) {
    const bool more = slab + 1 < CC;
    const char* const hb = halo0 + (slab & 1) * HALO;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) tap_body(tap / S, tap % S, slab, more, hb);
  }

  __syncthreads();
  constexpr int PITCH = BN * 2 + 64;
  static_assert(TH * 64 * PITCH <= 2 * HALO + 3 * B_STAGE, "epilogue tile fits the pipeline LDS");
  acc_tile_to_lds<FM, TN>(smem, PITCH, wm * 64, wn * TN * 32, n0, lane, acc, a.bias, a.Kout, a.act, a.slope);
  __syncthreads();
  const long long blk_base = a.out_base + n * a.out_sn + (long long)oh0 * a.out_sh + (long long)ow0 * a.out_sw;
  constexpr int VPR = BN / 8;
  for (int idx = tid; idx < TH * 64 * VPR; idx += 1024) {
    const int row = idx / VPR, v = idx - row * VPR;
    if (n0 + v * 8 >= a.Ks) continue;
    const long long off = blk_base + (long long)(row >> 6) * a.out_sh + (long long)(row & 63) * a.out_sw;
    *reinterpret_cast<u32x4*>(a.Y + off + n0 + v * 8) = *reinterpret_cast<const u32x4*>(smem + row * PITCH + v * 16);
  }
}

}  // namespace jpdse
