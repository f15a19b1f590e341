// Data gradient of the image head (ReflectionPad2d(3) + Conv2d(64 -> 3, 7x7), networks.py:148-152,243-246): dy has 3 (8 stored)
// channels, dx 64, both at full resolution.  The adjoint of the reflection is not a gather, so the gradient is computed on
// the padded domain (a 7x7 forward conv of the zero-extended dy with the flipped filter) and the 3-pixel ring is folded back.
// Round 1 / mid round 2: padded copy of dy + generic GEMM into a padded scratch tensor + reflect_fold_kernel over ALL of
// dx (reads 273 MB, writes 268 MB): 0.35 ms.
//
// Row-streaming form: filter in registers -- the single-phase data-gradient panel [c][r'][(s', k8)] (56 -> 64 elements per
// filter row, taps already flipped by the packer) is exactly the operand a dense 8-channel input needs: the run of output
// pixel q for filter row r' is the 128 contiguous bytes at pixel q of dy row oh - 6 + r' (8 pixels x 16 B; the 8th meets
// zero weights), i.e. 2 k-steps of v_mfma_f32_16x16x32_bf16 per filter row, A fragment = a plain 16-byte read at
// q * 16 + ks * 64 + 16 * (lane >> 4).  dy rows (1.1 KB per 64-pixel strip) stream through an LDS-DMA ring.
// The epilogue writes INTERIOR pixels of the padded domain straight into dx and only the ring pixels into the padded
// scratch tensor; reflect_ring_fold_kernel then adds the ring into the <= 2 % of dx pixels that have aliases.
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "gemm_halo.h"
#include "head_fwd.h"
#include "conv_rows.h"

namespace jpdse {

struct ThinInArgs {
  const bf16_t* DY;      // [N][H][W][8]           (input of the conv that is computed: dy for the head, the image for VGG conv1_1)
  const bf16_t* P;       // panel [64][R][32 KS]
  bf16_t* DX;            // DUAL: [N][H][W][64], interior of the padded domain; otherwise the output [N][OH][OW][64]
  bf16_t* DXP;           // DUAL: [N][H+6][W+6][64], ring pixels only
  const float* bias;     // forward use: bias + ReLU / LeakyReLU epilogue
  int act;
  float slope;
  int N, H, W;           // input size
  int OH, OW;            // output domain (DUAL: H + 6, W + 6)
  int py, px;            // input row = oh - py + r
  int TH, bands, strips;
};

// R x R filter over a dense 8-channel input, 64 outputs.  DUAL: the head data gradient (interior / ring destinations).
template <int RR> struct ThinInGeom {
  static constexpr int R = RR, PAD = 3;
  static constexpr int KS = (RR * 8 + 31) / 32;         // k32-steps per filter row: a run of 4 KS pixels (the last ones meet zero weights)
  static constexpr int PIX = 64 + 4 * KS - 1;           // staged pixels per row
  static constexpr int ROWB = 2048;                     // <= 71 x 16 B in two 1 KiB DMA units
  static constexpr int LA = 2, NR = R + LA + 1;         // R rows in use, LA in flight
  static constexpr int PITCH = 128 + 16;
  static constexpr int TILE = 64 * PITCH;
  static constexpr int LDS = NR * ROWB + TILE;
};

template <int RR, bool DUAL>
__global__ __launch_bounds__(256, 2) void thin_in_rows_kernel(const ThinInArgs a) {
  typedef ThinInGeom<RR> G;
  constexpr int KS = G::KS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wid & 1, wp = wid >> 1;
  const int HP = a.OH, WP = a.OW;
  int b = blockIdx.x;
  const int strip = b % a.strips; b /= a.strips;
  const int band = b % a.bands;
  const int n = b / a.bands;
  const int oh0 = band * a.TH, ow0 = strip * 64;        // padded-domain coordinates
  const int rows_here = HP - oh0 < a.TH ? HP - oh0 : a.TH;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const uint32_t smem0 = lds_addr32(smem);
  const uint32_t tile0 = smem0 + G::NR * G::ROWB;

  // ---- loader: waves 0 and 1 each move one 1 KiB unit (64 pixels x 16 B) of a dy row; pixel p of the strip = dy column
  // ow0 - 6 + p (zero outside the image)
  int col_off = -1;
  if (wid < 2) {
    const int p = wid * 64 + lane;
    const int iw = ow0 - a.px + p;
    if (p < G::PIX && (unsigned)iw < (unsigned)a.W) col_off = iw * 8;
  }
  const bf16_t* const img = a.DY + (long long)n * a.H * a.W * 8;
  const int row_elems = a.W * 8;
  auto issue_row = [&](int jr, int slot) {             // dy row oh0 - 6 + jr
    if (wid < 2) {
      const int ih = oh0 - a.py + jr;
      const bool row_ok = (unsigned)ih < (unsigned)a.H;
      const bf16_t* src = (row_ok && col_off >= 0) ? img + ih * (long long)row_elems + col_off : zero;
      glds16(src, smem + slot * G::ROWB + wid * 1024);
    }
  };
  constexpr int PRO = G::R + G::LA - 1;                 // rows of iterations 0 .. LA-1
#pragma unroll
  for (int jr = 0; jr < PRO; ++jr) issue_row(jr, jr);

  // ---- filter: breg[(r * 2 + ks) * 2 + j] = P[c = 32 wc + 16 j + (lane & 15)][r][32 ks + 8 (lane >> 4) ..]
  s16x8 breg[G::R * KS * 2];
  {
    const int kq = (lane >> 4) * 8;
#pragma unroll
    for (int r = 0; r < G::R; ++r)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int c = wc * 32 + j * 16 + (lane & 15);
          breg[(r * KS + ks) * 2 + j] = *reinterpret_cast<const s16x8*>(a.P + ((long long)c * G::R + r) * (32 * KS) + ks * 32 + kq);
        }
  }
#pragma unroll
  for (int t = 0; t < G::R * KS * 2; ++t) asm volatile("" : "+v"(breg[t]));
  float bv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    bv[j] = a.bias != nullptr ? a.bias[wc * 32 + j * 16 + (lane & 15)] : 0.f;
    asm volatile("" : "+v"(bv[j]));
  }
  const float nslope = a.act == JPDSE_ACT_RELU ? 0.f : (a.act == JPDSE_ACT_LRELU ? a.slope : 1.f);
  int a_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) a_off[i] = (wp * 32 + i * 16 + (lane & 15)) * 16 + (lane >> 4) * 16;
  const int odd = lane & 1;

  int base = 0, nslot = PRO % G::NR, njr = PRO;
  auto store_tile = [&](int oh) {                       // padded row oh: interior pixels -> dx, ring pixels -> the scratch tensor
    const bool row_in = !DUAL || (oh >= G::PAD && oh < a.H + G::PAD);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int idx = tid + 256 * k;
      const int px = idx >> 3, part = idx & 7;
      const int ow = ow0 + px;
      if (ow < WP) {
        const u32x4 val = *reinterpret_cast<const u32x4*>(smem + G::NR * G::ROWB + px * G::PITCH + part * 16);
        bf16_t* dst;
        if constexpr (DUAL) {
          const bool inner = row_in && ow >= G::PAD && ow < a.W + G::PAD;
          dst = inner ? a.DX + (((long long)n * a.H + (oh - G::PAD)) * a.W + (ow - G::PAD)) * 64
                      : a.DXP + (((long long)n * HP + oh) * WP + ow) * 64;
        } else {
          dst = a.DX + (((long long)n * HP + oh) * WP + ow) * 64;
        }
        *reinterpret_cast<u32x4*>(dst + part * 8) = val;
      }
    }
  };
  for (int i = 0; i < rows_here; ++i) {
    // conservative count (the tile stores of the last strip are partly masked): only the row DMAs issued for later
    // iterations may be in flight -- one instruction per row for waves 0 and 1, none for the others
    if (wid < 2) wait_vmcnt<G::LA - 1>(); else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();       // A: rows i .. i+6 complete; the tile of row i-1 is written
    asm volatile("" ::: "memory");
    if (i > 0) store_tile(oh0 + i - 1);
    issue_row(njr, nslot);
    ++njr;
    nslot = nslot + 1 == G::NR ? 0 : nslot + 1;

    f32x4 acc[2][2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i2][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int T = G::R * KS, DEPTH = T > 3 ? 3 : T - 1;
    s16x8 fr[DEPTH + 1][2];
    auto rd = [&](int t, s16x8 (&f)[2]) {
      const int r = t / KS, ks = t - KS * r;
      int slot = base + r;
      slot = slot >= G::NR ? slot - G::NR : slot;
      const uint32_t rb = smem0 + slot * G::ROWB + ks * 64;
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) f[i2] = lds_read128_asm(rb + a_off[i2]);
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
    base = base + 1 == G::NR ? 0 : base + 1;
    __builtin_amdgcn_s_barrier();       // B: the previous tile has been read (THINB)
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ep = 0; ep < 2; ++ep) {
          float v0 = acc[i2][j][2 * ep] + bv[j], v1 = acc[i2][j][2 * ep + 1] + bv[j];
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
  if (rows_here > 0) store_tile(oh0 + rows_here - 1);
}

// dx[h][w] += the ring aliases of (h, w) in the padded scratch tensor (adjoint of ReflectionPad2d(p), ring part only).
// One thread per 16-byte channel vector of a border pixel: the top / bottom p+1 rows, then the left / right p+1 columns of
// the rows between.
template <typename T>
__global__ void reflect_ring_fold_kernel(const T* __restrict__ dxp, T* __restrict__ dx, int N, int H, int W, int Cs, int p,
                                         long long total_vec) {
  constexpr int VE = 16 / sizeof(T);
  const int cv = Cs / VE;
  const int Hp = H + 2 * p, Wp = W + 2 * p;
  const int band = p + 1;
  const long long per_img = (long long)2 * band * W + (long long)(H - 2 * band) * 2 * band;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total_vec;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % cv);
    long long t = idx / cv;
    const int n = (int)(t / per_img);
    long long q = t % per_img;
    int h, w;
    if (q < (long long)2 * band * W) {
      const int rr = (int)(q / W);
      w = (int)(q % W);
      h = rr < band ? rr : H - 2 * band + rr;
    } else {
      q -= (long long)2 * band * W;
      const int rr = (int)(q / (2 * band)), cc = (int)(q % (2 * band));
      h = band + rr;
      w = cc < band ? cc : W - 2 * band + cc;
    }
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + p;
    if (h >= 1 && h <= p) hs[nh++] = p - h;
    if (h <= H - 2 && h >= H - 1 - p) hs[nh++] = p + 2 * (H - 1) - h;
    ws[nw++] = w + p;
    if (w >= 1 && w <= p) ws[nw++] = p - w;
    if (w <= W - 2 && w >= W - 1 - p) ws[nw++] = p + 2 * (W - 1) - w;
    if (nh == 1 && nw == 1) continue;
    T* const dst = dx + (((long long)n * H + h) * W + w) * Cs + c * VE;
    float accv[VE];
    Vec16<T>::load(dst, accv);
    for (int a = 0; a < nh; ++a)
      for (int b2 = 0; b2 < nw; ++b2) {
        if (a == 0 && b2 == 0) continue;            // the primary position was written into dx directly
        float v[VE];
        Vec16<T>::load(dxp + (((long long)n * Hp + hs[a]) * Wp + ws[b2]) * Cs + c * VE, v);
#pragma unroll
        for (int e = 0; e < VE; ++e) accv[e] += v[e];
      }
    Vec16<T>::store(dst, accv);
  }
}

}  // namespace jpdse
