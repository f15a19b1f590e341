// Image heads over 64-channel inputs at full resolution: ReflectionPad2d(3) + Conv2d(64 -> 3, 7x7) + Tanh
// (networks.py:148-152,243-246) and the data gradient of VGG19 conv1_1 (64 -> 3 channels, 3x3, zero padding).
// head_fwd_kernel factors the conv through all R*S taps (Z = x W, 147 columns) and then gathers 49 Z entries per
// output: per input row an MFMA phase, a 64 KB Z tile through LDS and an owner phase, serialised behind two barriers
// with one block per CU -- 0.37 ms where reading the input once takes 0.06.
//
// Row-streaming form (conv_rows.h's skeleton): the column taps go INTO the GEMM, the row taps into its columns:
//     Z'[q][(r, k)] = sum_{s, c} x[j][q + s][c] * w[k][r][s][c]        (N = R x 4 <= 28 columns, K-dim = S x 64)
//     y[j - r][q][k] += Z'[q][(r, k)]
// One MFMA pass per input row j with the A fragments read at s-shifted addresses (no im2col) and the 28-column filter in
// REGISTERS (R*4 k-steps x 4 VGPRs, loaded once per block); column (r, k) of the result belongs to output row j - r, so
// each lane adds its 16 pixels into that row of a small per-wave fp32 tile (read, add, write back) -- every address is
// touched by one lane per instruction and rows arrive in order, so the sum order is fixed (deterministic).  When row j - (R-1) is
// complete it is read back, biased, activated and stored (16 B per pixel), and its slot is zeroed for row j + 1.
// A block = 128 output pixels x TH rows, wave w = pixels [32w, 32w + 32).
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "gemm_halo.h"
#include "head_fwd.h"
#include "conv_rows.h"

namespace jpdse {

// CIN = 32 (round 4: the 32 -> 3 head of the LocalEnhancer, BASELINE config 3): 64-byte pixels, 16 per DMA unit, two k16-steps per
// column tap; the four 16-byte slots of a pixel are XOR-swizzled with (pixel >> 2) & 3, which keeps every 16-lane group of the
// fragment reads conflict free at any tap shift (pixels x, x+12, x+20, x+24 of a group share the bank quarter x & 3 and get the four
// different slot permutations).
template <int R, int CIN = 64> struct HeadRowsGeom {
  static constexpr int PIX = 128 + R - 1;
  static constexpr int PPU = 1024 / (CIN * 2);                // pixels per 1 KiB DMA unit: 8 (128-byte pixels) or 16 (64-byte pixels)
  static constexpr int UNITS = (PIX + PPU - 1) / PPU;         // 1 KiB DMA units per input row
  static constexpr int ROWB = UNITS * 1024;
  static constexpr int LA = 1, NR = LA + 2;                   // one row in use, LA in flight: 69 KB of LDS, two blocks per CU
  static constexpr int OROWS = R > 4 ? 8 : 4;                 // output rows under construction (power of two >= R)
  static constexpr int OPITCH = 32 * 16 + 16;                 // one output row of a wave: 32 pixels x 4 floats (+ bank skew)
  static constexpr int OTILE = OROWS * OPITCH;
  static constexpr int LDS = NR * ROWB + 4 * OTILE;
};

__device__ __forceinline__ void lds_add_f32(uint32_t addr, float v) {
  asm volatile("ds_add_f32 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

template <int R, int CIN = 64>
__global__ __launch_bounds__(256, 2) void head_rows_kernel(const HeadFwdArgs a, int TH, int bands, int strips) {
  static_assert(CIN == 64 || CIN == 32, "64- or 32-channel inputs");
  typedef HeadRowsGeom<R, CIN> G;
  constexpr int KPT = CIN / 16;                               // k16-steps per column tap
  constexpr int S = R, T = S * KPT;                           // k16-steps per input row
  constexpr int U0 = G::UNITS / 4, U1 = U0 + 1, EXTRA = G::UNITS % 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int strip = b % strips; b /= strips;
  const int band = b % bands;
  const int n = b / bands;
  const int oh0 = band * TH, ow0 = strip * 128;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const uint32_t smem0 = lds_addr32(smem);
  char* const otile = smem + G::NR * G::ROWB + wid * G::OTILE;
  const uint32_t otile0 = lds_addr32(otile);

  // zero the output rows under construction BEFORE any DMA is in flight (plain LDS stores: see head_fwd.h)
  for (int i = lane; i < G::OTILE / 16; i += 64) *reinterpret_cast<f32x4*>(otile + i * 16) = f32x4{0.f, 0.f, 0.f, 0.f};
  float fz = reinterpret_cast<const float*>(g_zero_page)[0];  // a zero the compiler cannot fold (MFMA -> asm LDS data: head_fwd.h)
  asm volatile("" : "+v"(fz));

  // ---- loader
  int col_off[U1];
#pragma unroll
  for (int k = 0; k < U1; ++k) {
    const int u = wid + 4 * k;
    const int lp = CIN == 64 ? u * 8 + (lane >> 3) : u * 16 + (lane >> 2);
    int iw = ow0 - a.pad + lp;
    bool ok = u < G::UNITS && lp < G::PIX;
    if (a.reflect) iw = iw < 0 ? -iw : (iw >= a.W ? 2 * (a.W - 1) - iw : iw);
    ok = ok && (unsigned)iw < (unsigned)a.W;
    const int chunk = CIN == 64 ? (((lane & 7) ^ (lp >> 1)) & 7) : (((lane & 3) ^ (lp >> 2)) & 3);
    col_off[k] = ok ? iw * CIN + chunk * 8 : -1;
  }
  const bf16_t* const ximg = a.X + (long long)n * a.H * a.W * CIN;
  const int row_elems = a.W * CIN;
  auto issue_row = [&](int jr, int slot) {
    int ih = oh0 + jr - a.pad;
    if (a.reflect) ih = ih < 0 ? -ih : (ih >= a.H ? 2 * (a.H - 1) - ih : ih);
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
#pragma unroll
  for (int jr = 0; jr <= G::LA; ++jr) issue_row(jr, jr);

  // ---- filter: column n = (r, k) = (n >> 2, n & 3), k-step t = s * 4 + ks: w[k][r][s][16 ks + 8 (lane >> 5) ..]
  s16x8 breg[T];
  {
    const int col = lane & 31, r = col >> 2, k = col & 3;
    const bool live = r < R && k < a.K;
    const bf16_t* const wp = live ? a.Wp + ((long long)(k * R + r) * S) * CIN + (lane >> 5) * 8 : zero;
#pragma unroll
    for (int t = 0; t < T; ++t) breg[t] = *reinterpret_cast<const s16x8*>(wp + (live ? ((t / KPT) * CIN + (t % KPT) * 16) : 0));
  }
#pragma unroll
  for (int t = 0; t < T; ++t) asm volatile("" : "+v"(breg[t]));
  float bias_v[3];                                            // loaded once: no memory instruction but DMA, LDS and stores in the loop
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    bias_v[k] = (a.bias != nullptr && k < a.K) ? a.bias[k] : 0.f;
    asm volatile("" : "+v"(bias_v[k]));
  }

  int a_base[S], a_sw[S];
#pragma unroll
  for (int s = 0; s < S; ++s) {
    const int lp = wid * 32 + (lane & 31) + s;
    a_base[s] = CIN == 64 ? lp << 7 : lp << 6;
    a_sw[s] = CIN == 64 ? ((lp >> 1) & 7) << 4 : ((lp >> 2) & 3) << 4;
  }
  const int hsel = lane >> 5;
  const int my_r = (lane & 31) >> 2, my_k = lane & 3;

  int base = 0, nslot = (G::LA + 1) % G::NR, njr = G::LA + 1;
  const int n_rows = TH + R - 1;
  for (int j = 0; j < n_rows; ++j) {
    {
      int k = j - (R - 1);
      k = k < 0 ? 0 : (k > G::LA ? G::LA : k);
      if (wid < EXTRA) wait_vmcnt_sel<G::LA * U1, 1, G::LA>(k); else wait_vmcnt_sel<G::LA * U0, 1, G::LA>(k);
    }
    __builtin_amdgcn_s_barrier();       // row j complete for every wave; the slot issued below held row j-1
    asm volatile("" ::: "memory");
    issue_row(njr, nslot);
    ++njr;
    nslot = nslot + 1 == G::NR ? 0 : nslot + 1;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const uint32_t rb = smem0 + base * G::ROWB;
    constexpr int DEPTH = 3;
    s16x8 fr[DEPTH + 1];
    auto rd = [&](int t) {
      const int s = t / KPT, ks = t % KPT;
      return lds_read128_asm(rb + a_base[s] + (((2 * ks + hsel) << 4) ^ a_sw[s]));
    };
#pragma unroll
    for (int t = 0; t < DEPTH; ++t) fr[t] = rd(t);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      if (t + DEPTH < T) fr[(t + DEPTH) % (DEPTH + 1)] = rd(t + DEPTH);
      s16x8& f = fr[t % (DEPTH + 1)];
      const int behind = (T - 1 - t) < DEPTH ? (T - 1 - t) : DEPTH;
      if (behind == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(f));
      else if (behind == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(f));
      else if (behind == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(f));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f));
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f, breg[t], acc, 0, 0, 0);
    }
    base = base + 1 == G::NR ? 0 : base + 1;

    // ---- column (r, k) -> output row j - r of the wave's tile: read, add, write back (ds_add_f32 is correct here too, but
    // LDS float atomics run lane by lane: 16 of them cost ~4 us per input row).  Each address belongs to one lane per
    // instruction, and the LDS executes one wave's instructions in order.
    {
      const int o = j - my_r;
      if (my_r < R && o >= 0 && o < TH) {
        const int ooff = (o & (G::OROWS - 1)) * G::OPITCH + my_k * 4;
        float cur[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) cur[e] = *reinterpret_cast<const float*>(otile + ooff + ((e & 3) + 8 * (e >> 2) + 4 * hsel) * 16);
#pragma unroll
        for (int e = 0; e < 16; ++e) lds_store32(otile0 + ooff + ((e & 3) + 8 * (e >> 2) + 4 * hsel) * 16, cur[e] + (acc[e] + fz));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- output row j - (R-1) is complete
    const int of = j - (R - 1);
    if (of >= 0) {
      char* const orow = otile + (of & (G::OROWS - 1)) * G::OPITCH;
      if (lane < 32) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(orow + lane * 16);
        float o8[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o8[k] = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < a.K) o8[k] = apply_act(v[k] + bias_v[k], a.act, a.slope);
        Vec16<bf16_t>::store(a.Y + (((long long)n * a.OH + oh0 + of) * a.OW + ow0 + wid * 32 + lane) * a.Ks_out, o8);
        asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr32(orow + lane * 16)), "v"(f32x4{fz, fz, fz, fz}) : "memory");
      }
    }
  }
}

}  // namespace jpdse
