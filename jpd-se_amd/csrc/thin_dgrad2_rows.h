// Data gradient of PatchGAN layer 0 (Conv2d 4x4 stride 2 pad 2, 64 outputs) with respect to <= 3 input channels (the image
// part of the discriminator input: HipConv2d.bwd_input_slice): dy has 64 channels at 257x513, dx 3 (8 stored) at 512x1024.
// As four stride-phase GEMMs on the generic kernel (padded copy of dy, 3 live columns of 32) it took 0.20 ms where reading
// dy once takes 0.02.
//
// head_rows.h's scheme for a transposed stride-2 conv: one MFMA pass per dy row j, output pixels in pairs m' (columns
// 2m', 2m'+1), the two column taps of a parity in the K-dim (shifted A reads), row tap and column parity in the GEMM columns:
//     Z'[m'][(r, b, k)] = sum_{t, c} dy[j][m' + 1 - t][c] * w[c][r][b + 2t][k]          (N = 4 x 2 x 4 = 32, K-dim = 2 x 64)
//     dx[2j - 2 + r][2m' + b][k] += Z'[m'][(r, b, k)]
// Every dx row receives two contributions (dy rows j and j-1); rows 2j-2 and 2j-1 are complete after dy row j and leave
// as 16-byte pixels.  Filter in registers (8 k-steps), dy rows through an LDS-DMA ring, per-wave fp32 row tiles updated by
// read-add-write in a fixed order (deterministic), as in head_rows.h.  Block = 256 output pixels x TH rows, wave = 64 pixels.
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "gemm_halo.h"
#include "head_fwd.h"
#include "conv_rows.h"

namespace jpdse {

struct ThinDgrad2Args {
  const bf16_t* DY;      // [N][OH][OW][64]
  const bf16_t* P[4];    // stride-phase panels of the sliced filter, plan order (qh, qw); rows = 8 stored output channels
  bf16_t* DX;            // [N][H][W][8]
  int N, OH, OW, H, W, K;
  int TH, bands, strips;
};

struct ThinDgrad2Geom {
  static constexpr int PIX = 129;
  static constexpr int UNITS = (PIX + 7) / 8;
  static constexpr int ROWB = UNITS * 1024;
  static constexpr int LA = 1, NR = LA + 2;            // 69 KB of LDS: two blocks per CU
  static constexpr int OPITCH = 64 * 16 + 16;          // one dx row of a wave: 64 pixels x 4 floats (+ bank skew)
  static constexpr int OTILE = 4 * OPITCH;             // dx rows under construction
  static constexpr int LDS = NR * ROWB + 4 * OTILE;
};

__global__ __launch_bounds__(256, 2) void thin_dgrad2_rows_kernel(const ThinDgrad2Args a) {
  typedef ThinDgrad2Geom G;
  constexpr int T = 8;                                 // k16-steps per dy row: 2 column taps x 64 channels
  constexpr int U0 = G::UNITS / 4, U1 = U0 + 1, EXTRA = G::UNITS % 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int strip = b % a.strips; b /= a.strips;
  const int band = b % a.bands;
  const int n = b / a.bands;
  const int o0 = band * a.TH, m0 = strip * 128;        // first dx row (even), first pixel pair
  const int j0 = o0 / 2;                               // dy row of iteration 0
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const uint32_t smem0 = lds_addr32(smem);
  char* const otile = smem + G::NR * G::ROWB + wid * G::OTILE;
  const uint32_t otile0 = lds_addr32(otile);
  for (int i = lane; i < G::OTILE / 16; i += 64) *reinterpret_cast<f32x4*>(otile + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  float fz = reinterpret_cast<const float*>(g_zero_page)[0];
  asm volatile("" : "+v"(fz));

  int col_off[U1];
#pragma unroll
  for (int k = 0; k < U1; ++k) {
    const int u = wid + 4 * k;
    const int lp = u * 8 + (lane >> 3);                // LDS pixel = dy column m0 + lp
    const int ow = m0 + lp;
    const bool ok = u < G::UNITS && lp < G::PIX && ow < a.OW;
    const int chunk = ((lane & 7) ^ (lp >> 1)) & 7;
    col_off[k] = ok ? ow * 64 + chunk * 8 : -1;
  }
  const bf16_t* const img = a.DY + (long long)n * a.OH * a.OW * 64;
  const int row_elems = a.OW * 64;
  auto issue_row = [&](int jr, int slot) {
    const int oh = j0 + jr;
    const bool row_ok = oh < a.OH;
    const bf16_t* const xrow = img + (row_ok ? oh : 0) * (long long)row_elems;
    char* const dst = smem + slot * G::ROWB;
#pragma unroll
    for (int k = 0; k < U1; ++k) {
      if (k < U0 || wid < EXTRA) {
        const bf16_t* src = (row_ok && col_off[k] >= 0) ? xrow + col_off[k] : zero;
        glds16(src, dst + (wid + 4 * k) * 1024);
      }
    }
  };
#pragma unroll
  for (int jr = 0; jr <= G::LA; ++jr) issue_row(jr, jr);

  // ---- filter: column n = (r, b, k) = (n >> 3, (n >> 2) & 1, n & 3), k-step (t, ks): w[c][r][b + 2t][k] =
  // panel(qh = r & 1, qw = b)[k][up = 1 - (r >> 1)][wp = 1 - t][c]   (rows of 2 x 2 x 64 elements)
  s16x8 breg[T];
  {
    const int col = lane & 31, r = col >> 3, bb = (col >> 2) & 1, k = col & 3;
    const bool live = k < a.K;
    const bf16_t* const pan = a.P[(r & 1) * 2 + bb];
#pragma unroll
    for (int t2 = 0; t2 < T; ++t2) {
      const int t = t2 >> 2, ks = t2 & 3;
      const bf16_t* src = live ? pan + (long long)k * 256 + ((1 - (r >> 1)) * 2 + (1 - t)) * 64 + ks * 16 + (lane >> 5) * 8 : zero;
      breg[t2] = *reinterpret_cast<const s16x8*>(src);
    }
  }
#pragma unroll
  for (int t2 = 0; t2 < T; ++t2) asm volatile("" : "+v"(breg[t2]));

  int a_base[2], a_sw[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int lp = wid * 32 + (lane & 31) + 1 - t;
    a_base[t] = lp << 7;
    a_sw[t] = ((lp >> 1) & 7) << 4;
  }
  const int hsel = lane >> 5;
  const int my_r = (lane & 31) >> 3, my_b = ((lane & 31) >> 2) & 1, my_k = lane & 3;

  int base = 0, nslot = (G::LA + 1) % G::NR, njr = G::LA + 1;
  const int n_it = a.TH / 2 + 1;
  for (int jj = 0; jj < n_it; ++jj) {
    {
      int k = jj - 1;
      k = k < 0 ? 0 : (k > G::LA ? G::LA : k);
      if (wid < EXTRA) wait_vmcnt_sel<G::LA * U1, 2, G::LA>(k); else wait_vmcnt_sel<G::LA * U0, 2, G::LA>(k);
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    issue_row(njr, nslot);
    ++njr;
    nslot = nslot + 1 == G::NR ? 0 : nslot + 1;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const uint32_t rb = smem0 + base * G::ROWB;
    s16x8 fr[T];
#pragma unroll
    for (int t2 = 0; t2 < T; ++t2) fr[t2] = lds_read128_asm(rb + a_base[t2 >> 2] + (((2 * (t2 & 3) + hsel) << 4) ^ a_sw[t2 >> 2]));
#pragma unroll
    for (int t2 = 0; t2 < T; ++t2) {
      if (t2 == 0) asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(fr[0]));
      else if (t2 == 1) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(fr[1]));
      else if (t2 == 2) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fr[2]));
      else if (t2 == 3) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(fr[3]));
      else if (t2 == 4) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fr[4]));
      else if (t2 == 5) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fr[5]));
      else if (t2 == 6) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fr[6]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr[7]));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fr[t2], breg[t2], acc, 0, 0, 0);
    }
    base = base + 1 == G::NR ? 0 : base + 1;

    // ---- column (r, b, k) -> dx row 2 jj - 2 + r of the band, pixel 2 m' + b
    {
      const int orel = 2 * jj - 2 + my_r;
      if (orel >= 0 && orel < a.TH) {
        const int ooff = (orel & 3) * G::OPITCH + my_b * 16 + my_k * 4;
        float cur[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[e] = *reinterpret_cast<const float*>(otile + ooff + ((e & 3) + 8 * (e >> 2) + 4 * hsel) * 32);
#pragma unroll
        for (int e = 0; e < 16; ++e) lds_store32(otile0 + ooff + ((e & 3) + 8 * (e >> 2) + 4 * hsel) * 32, cur[e] + (acc[e] + fz));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- dx rows 2 jj - 2 and 2 jj - 1 are complete: 64 pixels x 16 B each
    if (jj >= 1) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int orel = 2 * jj - 2 + q;
        char* const orow = otile + (orel & 3) * G::OPITCH;
        const f32x4 v = *reinterpret_cast<const f32x4*>(orow + lane * 16);
        float o8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o8[k] = (k < 3 && k < a.K) ? v[k] : 0.f;
        Vec16<bf16_t>::store(a.DX + (((long long)n * a.H + o0 + orel) * a.W + 2 * (m0 + wid * 32) + lane) * 8, o8);
        asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr32(orow + lane * 16)), "v"(f32x4{fz, fz, fz, fz}) : "memory");
      }
    }
  }
}

}  // namespace jpdse
