// Forward of PatchGAN layer 0 (Conv2d(39 -> 64, 4x4, stride 2, pad 2) + LeakyReLU, networks.py:430) over the batched
// [label|fake ; label|real] input at full resolution: 8 x 512 x 1024 x 40 in, 8 x 257 x 513 x 64 out.  thin_fwd_kernel
// stages the strips of a 4 x 64 output tile once and streams the four filter rows through a barrier-stepped ring: short
// per-tile pipelines, 0.23 ms (360 TFLOP/s) where the bytes take 0.1.
//
// Row-streaming form (conv_rows.h): filter in REGISTERS, input rows streamed once through an LDS-DMA ring, one output row
// per iteration.  The input stays dense ([pixel][40 channels], 80 B per pixel): the run of output pixel q for filter row r is the
// 320 contiguous bytes at pixel 2q of input row 2 oh + r - 2, i.e. K-dim 160 = 5 k-steps of v_mfma_f32_16x16x32_bf16 per
// filter row, and the A fragment of (q, k-step) is a plain 16-byte read at q * 160 + ks * 64 + 16 * (lane >> 4) -- conflict
// free without any swizzle (the eight even 16-byte quads for the lanes of one k-chunk, the odd ones for the next).
// Block = 64 output pixels x TH rows; wave (wc, wp) = 32 channels (two 16-column MFMA tiles) x 32 pixels (two 16-row tiles);
// B = 4 rows x 5 k-steps x 2 tiles x 4 VGPRs = 160 VGPRs, from the thin panel [R][K][KP] the packer already writes.
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "gemm_halo.h"
#include "head_fwd.h"
#include "conv_rows.h"
#include "thin_fwd.h"

namespace jpdse {

struct ThinRowsGeom {
  static constexpr int PIX = 130;                       // staged input pixels per row: 2 * 63 + 4
  static constexpr int UNITS = (PIX * 80 + 1023) / 1024;    // dense bytes in 1 KiB DMA units
  static constexpr int ROWB = UNITS * 1024;
  static constexpr int LA = 1, NR = 6;                  // 4 rows in use, 2 * LA in flight: 75 KB of LDS, two blocks per CU (each
                                                        // covers the other's filter load, prologue and row latencies)
  static constexpr int PITCH = 128 + 16;                // output tile: 64 pixels x (64 channels bf16 + pad)
  static constexpr int TILE = 64 * PITCH;
  static constexpr int LDS = NR * ROWB + TILE;
};

__global__ __launch_bounds__(256, 2) void thin_rows_kernel(const ThinFwdArgs a, int TH, int bands) {
  typedef ThinRowsGeom G;
  constexpr int U0 = G::UNITS / 4, U1 = U0 + 1, EXTRA = G::UNITS % 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wid & 1, wp = wid >> 1;
  int b = blockIdx.x;
  const int strip = b % a.tiles_w; b /= a.tiles_w;
  const int band = b % bands;
  const int n = b / bands;
  const int oh0 = band * TH, ow0 = strip * 64;
  const int rows_here = a.OH - oh0 < TH ? a.OH - oh0 : TH;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const uint32_t smem0 = lds_addr32(smem);
  const uint32_t tile0 = smem0 + G::NR * G::ROWB;

  // ---- loader: 16-byte chunk g of the dense strip = pixel g / 5, channel part g % 5
  int col_off[U1];
#pragma unroll
  for (int k = 0; k < U1; ++k) {
    const int u = wid + 4 * k;
    const int g = u * 64 + lane;
    const int px = g / 5, part = g - px * 5;
    const int iw = ow0 * 2 - a.pad + px;
    const bool ok = u < G::UNITS && px < G::PIX && (unsigned)iw < (unsigned)a.W;
    col_off[k] = ok ? iw * 40 + part * 8 : -1;
  }
  const bf16_t* const ximg = a.X + (long long)n * a.H * a.W * 40;
  const int row_elems = a.W * 40;
  const int ih_base = oh0 * 2 - a.pad;
  auto issue_row = [&](int jr, int slot) {
    const int ih = ih_base + jr;
    const bool row_ok = (unsigned)ih < (unsigned)a.H;
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
  constexpr int PRO = 2 * (G::LA - 1) + 4;
#pragma unroll
  for (int jr = 0; jr < PRO; ++jr) issue_row(jr, jr);

  // ---- filter: breg[(r * 5 + ks) * 2 + j] = panel[r][k = 32 wc + 16 j + (lane & 15)][32 ks + 8 (lane >> 4) ..]
  s16x8 breg[40];
  {
    const int kq = (lane >> 4) * 8;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ks = 0; ks < 5; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = wc * 32 + j * 16 + (lane & 15);
          breg[(r * 5 + ks) * 2 + j] = *reinterpret_cast<const s16x8*>(a.Wt + ((long long)r * a.K + k) * a.KP + ks * 32 + kq);
        }
  }
#pragma unroll
  for (int t = 0; t < 40; ++t) asm volatile("" : "+v"(breg[t]));
  float bv[2][1];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bv[j][0] = a.bias != nullptr ? a.bias[wc * 32 + j * 16 + (lane & 15)] : 0.f;
    asm volatile("" : "+v"(bv[j][0]));
  }
  const float nslope = a.act == JPDSE_ACT_RELU ? 0.f : (a.act == JPDSE_ACT_LRELU ? a.slope : 1.f);

  // A addressing: pixel q = 32 wp + 16 i + (lane & 15), k-chunk lane >> 4
  int a_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_off[i] = (wp * 32 + i * 16 + (lane & 15)) * 160 + (lane >> 4) * 16;
  const int odd = lane & 1;

  int base = 0, nslot = PRO % G::NR, njr = PRO;
  for (int i = 0; i < rows_here; ++i) {
    // conservative count: only the row DMAs of later iterations may be in flight (the tile stores of the last strip are
    // partly masked, so their number is not a constant)
    if (wid < EXTRA) wait_vmcnt<(G::LA - 1) * 2 * U1>(); else wait_vmcnt<(G::LA - 1) * 2 * U0>();
    __builtin_amdgcn_s_barrier();       // A: the four rows of this iteration are complete; the tile of row i-1 is written
    asm volatile("" ::: "memory");
    if (i > 0) {
      const long long orow = (((long long)n * a.OH + oh0 + i - 1) * a.OW + ow0) * 64;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = tid + 256 * k;
        const int px = idx >> 3, part = idx & 7;
        if (ow0 + px < a.OW)
          *reinterpret_cast<u32x4*>(a.Y + orow + px * 64 + part * 8) =
              *reinterpret_cast<const u32x4*>(smem + G::NR * G::ROWB + px * G::PITCH + part * 16);
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      issue_row(njr, nslot);
      ++njr;
      nslot = nslot + 1 == G::NR ? 0 : nslot + 1;
    }

    f32x4 acc[2][2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i2][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t rbase[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int slot = base + r;
      slot = slot >= G::NR ? slot - G::NR : slot;
      rbase[r] = smem0 + slot * G::ROWB;
    }
    constexpr int T = 20, DEPTH = 3;                    // (filter row, k-step) units; units of fragments in flight
    s16x8 fr[DEPTH + 1][2];
    auto rd = [&](int t, s16x8 (&f)[2]) {
      const int r = t / 5, ks = t - 5 * r;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) f[i2] = lds_read128_asm(rbase[r] + a_off[i2] + ks * 64);
    };
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) rd(t, fr[t]);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + DEPTH < T) rd(t + DEPTH, fr[(t + DEPTH) % (DEPTH + 1)]);
      s16x8 (&f)[2] = fr[t % (DEPTH + 1)];
      const int behind = (T - 1 - t) < DEPTH ? (T - 1 - t) : DEPTH;
      if (behind == 3) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(f[0]), "+v"(f[1]));
      else if (behind == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0]), "+v"(f[1]));
      else if (behind == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f[0]), "+v"(f[1]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]));
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i2][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[i2], breg[t * 2 + j], acc[i2][j], 0, 0, 0);
    }
    base += 2;
    base = base >= G::NR ? base - G::NR : base;
    __builtin_amdgcn_s_barrier();       // B: the tile of row i-1 has been read by every thread
    asm volatile("" ::: "memory");
    // ---- accumulators (16x16: column lane & 15, rows 4 (lane >> 4) + e) -> bias, activation, bf16 -> tile[pixel][channel]
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ep = 0; ep < 2; ++ep) {
          float v0 = acc[i2][j][2 * ep] + bv[j][0], v1 = acc[i2][j][2 * ep + 1] + bv[j][0];
          v0 = v0 > 0.f ? v0 : v0 * nslope;
          v1 = v1 > 0.f ? v1 : v1 * nslope;
          const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, odd ? v0 : v1), 0xB1, 0xF, 0xF, false));
          const float lo = odd ? recv : v0, hi = odd ? v1 : recv;
          const uint32_t word = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
          const int px = wp * 32 + i2 * 16 + 4 * (lane >> 4) + 2 * ep + odd;
          const int ch = wc * 32 + j * 16 + (lane & 15) - odd;
          lds_store32u(tile0 + px * G::PITCH + ch * 2, word);
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (rows_here > 0) {
    const long long orow = (((long long)n * a.OH + oh0 + rows_here - 1) * a.OW + ow0) * 64;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + 256 * k;
      const int px = idx >> 3, part = idx & 7;
      if (ow0 + px < a.OW)
        *reinterpret_cast<u32x4*>(a.Y + orow + px * 64 + part * 8) =
            *reinterpret_cast<const u32x4*>(smem + G::NR * G::ROWB + px * G::PITCH + part * 16);
    }
  }
}

}  // namespace jpdse
