// Data gradient of the 3x3 stride-2 pad-1 convolution 64 -> 128 at full resolution (G's first down-sampling conv, and --
// the same operator -- the forward of the last ConvTranspose2d 128 -> 64, networks.py:215,244): dy has 128 channels at
// 256x512, dx 64 channels at 512x1024.  As four stride-phase GEMMs on the tiled kernel it runs at 310 TFLOP/s: K-dims
// of 128..512 (2..8 K-tiles) never fill a pipeline, dy is re-staged once per phase and tap, the filter once per tile.
//
// conv_rows.h's scheme, transposed: the FILTER LIVES IN REGISTERS, dy streams through LDS once, and one pass over a dy
// row produces BOTH output rows it feeds, all four phases at once:
//   dx[2a  ][2b  ] = dy[a][b] W11                                   (phase qh=1,qw=1: one tap)
//   dx[2a  ][2b+1] = dy[a][b] W10' + dy[a][b+1] W10''               (two taps)
//   dx[2a+1][2b  ] = dy[a][b] W01' + dy[a+1][b] W01''               (two taps)
//   dx[2a+1][2b+1] = dy[a][b] W00' + dy[a][b+1] .. + dy[a+1][b] .. + dy[a+1][b+1] ..   (four taps)
// (W.. = the stride-phase panels of the packed data-gradient filter, taps already flipped by the packer.)
// A block owns 64 dy columns x TH dy rows; wave w owns output channels [16w, 16w+16) for all phases: 9 taps x 4 k-steps of
// v_mfma_f32_16x16x32_bf16 B fragments = 144 VGPRs, loaded once.  Per dy row and 16-pixel m-tile the four A fragments
// (dy[a|a+1][b|b+1]) feed 9 MFMAs per k-step; reads are inline asm, two (m-tile, k-step) units ahead, counted lgkmcnt.
// The 2 x 128 output pixels of a dy row go through one LDS tile and leave as full 128-byte pixels (8 lanes x 16 B).
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "gemm_halo.h"
#include "head_fwd.h"
#include "conv_rows.h"

namespace jpdse {

struct Dgrad2Args {
  const bf16_t* DY;      // [N][OH][OW][128]
  const bf16_t* P[4];    // phase panels, plan order (qh, qw) = (0,0), (0,1), (1,0), (1,1); rows = 64 output channels
  bf16_t* DX;            // [N][2 OH][2 OW][64]
  int N, OH, OW;
  int TH, bands, strips;
  float* mom;            // optional moments of dx for the InstanceNorm that follows: [N][64][mom_slots][2], slot = band * strips + strip
  int mom_slots;
};

struct Dgrad2Geom {
  static constexpr int PIX = 65;                       // staged dy pixels per row (64 + the right neighbour)
  static constexpr int UNITS = (PIX * 256 + 1023) / 1024;   // 1 KiB DMA units: 4 pixels x 256 B
  static constexpr int ROWB = UNITS * 1024;
  static constexpr int LA = 5, NR = LA + 2;            // rows in flight ahead of the two in use
  // output tile: [2][128 pixels] x 64 channels bf16, 128-byte rows whose eight 16-byte parts are XOR-swizzled with bits 1 and 3
  // of the row index (round 4): the 16-lane groups of the tile's ds_read_b128 (pixel p parts 0-3 / 4-7 beside pixels p+1 .. p+3)
  // and the 32-lane groups of its ds_write_b32 (rows r, r+2, r+8, r+10 of one channel pair) are then conflict free.  With the
  // padded 144-byte rows of rounds 2-3 the reads conflicted two ways (SQ_LDS_BANK_CONFLICT 39 % of SQ_LDS_IDX_ACTIVE).
  static constexpr int PITCH = 128;
  static constexpr int TILE = 2 * 128 * PITCH;
  static constexpr int LDS = NR * ROWB + TILE;
};

__device__ __forceinline__ int d2_tile_off(int row, int part) {       // byte offset of 16-byte part `part` of tile row `row`
  return (row << 7) + ((part ^ (((row >> 1) & 1) | (((row >> 3) & 1) << 1))) << 4);
}

// NOCONF (developer build, TIMING ONLY -- the results are wrong): every LDS access that can have a bank conflict is re-pointed at a
// conflict-free address -- the column-shifted A-fragment reads at the unshifted pixel, the tile reads lane-linear -- to measure what
// the kernel would gain from a conflict-free layout (profiles/r04_dgrad2_rows_conflicts_ab.txt).
template <bool NOCONF = false>
__global__ __launch_bounds__(256) void dgrad2_rows_kernel(const Dgrad2Args a) {
  typedef Dgrad2Geom G;
  constexpr int U0 = G::UNITS / 4, U1 = U0 + 1, EXTRA = G::UNITS % 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int b = blockIdx.x;
  const int strip = b % a.strips; b /= a.strips;
  const int band = b % a.bands;
  const int n = b / a.bands;
  const int a0 = band * a.TH, b0 = strip * 64;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const uint32_t smem0 = lds_addr32(smem);
  const uint32_t tile0 = smem0 + G::NR * G::ROWB;

  // ---- loader: this wave's units of a dy row (u = wid, wid + 4, ...): 4 pixels x 16 chunks of 16 B per unit
  int col_off[U1];
#pragma unroll
  for (int k = 0; k < U1; ++k) {
    const int u = wid + 4 * k;
    const int lp = u * 4 + (lane >> 4);                // LDS pixel = dy column b0 + lp
    const int bw = b0 + lp;
    const bool ok = u < G::UNITS && lp < G::PIX && bw < a.OW;
    // 16-byte chunk c of pixel p sits in slot c ^ ((p & 7) << 1): conflict-free for the 16x16x32 A-fragment reads at BOTH column
    // shifts (pixel and pixel + 1) in every 16-lane group of ds_read_b128 -- found by exhaustive search over the XOR-linear maps
    // (round 4).  Rounds 2-3 used c ^ (p & 15): free of conflicts at shift 0, two colliding lanes per group at shift 1 (SQ_LDS_BANK_CONFLICT
    // 39 % of SQ_LDS_IDX_ACTIVE together with the output tile; profiles/r04_dgrad2_rows_conflicts_ab.txt).
    const int chunk = ((lane & 15) ^ ((lp & 7) << 1)) & 15;
    col_off[k] = ok ? bw * 128 + chunk * 8 : -1;
  }
  const bf16_t* const img = a.DY + (long long)n * a.OH * a.OW * 128;
  const int row_elems = a.OW * 128;
  auto issue_row = [&](int jr, int slot) {             // dy row a0 + jr -> ring slot
    const int ih = a0 + jr;
    const bool row_ok = ih < a.OH;
    const bf16_t* const xrow = img + (row_ok ? ih : 0) * (long long)row_elems;
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

  // ---- filter slice: B[t][ks], t = 0: ee | 1, 2: eo | 3, 4: oe | 5..8: oo (order of use below)
  s16x8 breg[36];
  {
    const int c = wid * 16 + (lane & 15);              // output channel (row of the panels)
    const int kq = (lane >> 4) * 8;
    // phase (qh, qw): Uh = qh ? 1 : 2, Uw = qw ? 1 : 2; element [c][up][wp][k] at c * Uh * Uw * 128 + (up * Uw + wp) * 128 + k
    const bf16_t* const p00 = a.P[0] + (long long)c * 512 + kq;   // (0,0): 2 x 2 taps
    const bf16_t* const p01 = a.P[1] + (long long)c * 256 + kq;   // (0,1): 2 x 1
    const bf16_t* const p10 = a.P[2] + (long long)c * 256 + kq;   // (1,0): 1 x 2
    const bf16_t* const p11 = a.P[3] + (long long)c * 128 + kq;   // (1,1): 1 x 1
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      breg[0 * 4 + ks] = *reinterpret_cast<const s16x8*>(p11 + ks * 32);
      breg[1 * 4 + ks] = *reinterpret_cast<const s16x8*>(p10 + 0 * 128 + ks * 32);     // wp = 0: dy[a][b]
      breg[2 * 4 + ks] = *reinterpret_cast<const s16x8*>(p10 + 1 * 128 + ks * 32);     // wp = 1: dy[a][b+1]
      breg[3 * 4 + ks] = *reinterpret_cast<const s16x8*>(p01 + 0 * 128 + ks * 32);     // up = 0: dy[a][b]
      breg[4 * 4 + ks] = *reinterpret_cast<const s16x8*>(p01 + 1 * 128 + ks * 32);     // up = 1: dy[a+1][b]
      breg[5 * 4 + ks] = *reinterpret_cast<const s16x8*>(p00 + 0 * 128 + ks * 32);     // (0,0): dy[a][b]
      breg[6 * 4 + ks] = *reinterpret_cast<const s16x8*>(p00 + 1 * 128 + ks * 32);     // (0,1): dy[a][b+1]
      breg[7 * 4 + ks] = *reinterpret_cast<const s16x8*>(p00 + 2 * 128 + ks * 32);     // (1,0): dy[a+1][b]
      breg[8 * 4 + ks] = *reinterpret_cast<const s16x8*>(p00 + 3 * 128 + ks * 32);     // (1,1): dy[a+1][b+1]
    }
  }
#pragma unroll
  for (int t = 0; t < 36; ++t) asm volatile("" : "+v"(breg[t]));

  // A fragment addressing (16x16x32: lane -> pixel lane & 15, 8 k-values at 8 * (lane >> 4)): byte offset inside a ring row
  // of pixel px, k-step ks: px * 256 + ((ks * 4 + (lane >> 4)) ^ ((px & 7) << 1)) * 16
  int a_px[4][2];                                       // [m-tile][column shift]
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int sh = 0; sh < 2; ++sh) a_px[mt][sh] = mt * 16 + (lane & 15) + sh;
  const int kc = lane >> 4;
  const int odd = lane & 1;
  const long long img_out = (long long)n * (2 * a.OH) * (2 * a.OW) * 64;

  float ms1 = 0.f, ms2 = 0.f, mpilot = 0.f;             // moments about a pilot (common.h)
  int base = 0;                                         // ring slot of dy row a0 + i
  int nslot = (G::LA + 1) % G::NR, njr = G::LA + 1;
  for (int i = 0; i < a.TH; ++i) {
    // row i+1 landed (row i landed an iteration earlier); behind it: LA-1 rows and the tile stores of up to LA-1 iterations
    {
      int k = i - 1;
      k = k < 0 ? 0 : (k > G::LA - 1 ? G::LA - 1 : k);
      if (wid < EXTRA) wait_vmcnt_sel<(G::LA - 1) * U1, 8, G::LA - 1>(k); else wait_vmcnt_sel<(G::LA - 1) * U0, 8, G::LA - 1>(k);
    }
    __builtin_amdgcn_s_barrier();       // A: rows i, i+1 complete for every wave; the tile of iteration i-1 is written
    asm volatile("" ::: "memory");
    if (i > 0) {
      // ---- tile of dy row i-1 -> dx rows 2(a0+i-1), +1: 256 pixels x 128 B, 8 x 16 B per thread
      const long long orow = img_out + (long long)(2 * (a0 + i - 1)) * (2 * a.OW) * 64 + (long long)(2 * b0) * 64;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = tid + 256 * k;
        const int pxl = idx >> 3, part = idx & 7;
        const u32x4 val = *reinterpret_cast<const u32x4*>(smem + G::NR * G::ROWB + (NOCONF ? idx * 16 : d2_tile_off(pxl, part)));
        *reinterpret_cast<u32x4*>(a.DX + orow + (long long)(pxl >> 7) * (2 * a.OW) * 64 + (pxl & 127) * 64 + part * 8) = val;
      }
    }
    issue_row(njr, nslot);
    ++njr;
    nslot = nslot + 1 == G::NR ? 0 : nslot + 1;

    f32x4 acc[4][4];                                    // [m-tile][ee, eo, oe, oo]
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) acc[mt][ph] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t r0 = smem0 + base * G::ROWB;
    const uint32_t r1 = smem0 + (base + 1 == G::NR ? 0 : base + 1) * G::ROWB;
    s16x8 fr[3][4];                                     // [unit in flight][dy[a][b], dy[a][b+1], dy[a+1][b], dy[a+1][b+1]]
    auto rd = [&](int u, s16x8 (&f)[4]) {               // unit u = mt * 4 + ks
      const int mt = u >> 2, ks = u & 3;
      const int c = ks * 4 + kc;
      const uint32_t o0 = (a_px[mt][0] << 8) + (((c ^ ((a_px[mt][0] & 7) << 1)) & 15) << 4);
      const uint32_t o1 = NOCONF ? o0 : (a_px[mt][1] << 8) + (((c ^ ((a_px[mt][1] & 7) << 1)) & 15) << 4);
      f[0] = lds_read128_asm(r0 + o0);
      f[1] = lds_read128_asm(r0 + o1);
      f[2] = lds_read128_asm(r1 + o0);
      f[3] = lds_read128_asm(r1 + o1);
    };
    rd(0, fr[0]);
    rd(1, fr[1]);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (u + 2 < 16) rd(u + 2, fr[(u + 2) % 3]);
      s16x8 (&f)[4] = fr[u % 3];
      if (u + 2 < 16) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
      else if (u + 1 < 16) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]));
      const int mt = u >> 2, ks = u & 3;
      acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], breg[0 * 4 + ks], acc[mt][0], 0, 0, 0);
      acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], breg[1 * 4 + ks], acc[mt][1], 0, 0, 0);
      acc[mt][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], breg[3 * 4 + ks], acc[mt][2], 0, 0, 0);
      acc[mt][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[0], breg[5 * 4 + ks], acc[mt][3], 0, 0, 0);
      acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], breg[2 * 4 + ks], acc[mt][1], 0, 0, 0);
      acc[mt][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[1], breg[6 * 4 + ks], acc[mt][3], 0, 0, 0);
      acc[mt][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[2], breg[4 * 4 + ks], acc[mt][2], 0, 0, 0);
      acc[mt][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[2], breg[7 * 4 + ks], acc[mt][3], 0, 0, 0);
      acc[mt][3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[3], breg[8 * 4 + ks], acc[mt][3], 0, 0, 0);
    }
    base = base + 1 == G::NR ? 0 : base + 1;
    if (a.mom != nullptr) {
      if (i == 0) mpilot = bf16_round(acc[0][0][0]);      // this lane's own pilot (no cross-lane traffic in this loop)
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int ph = 0; ph < 4; ++ph)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float v = bf16_round(acc[mt][ph][e]) - mpilot;
            ms1 += v;
            ms2 += v * v;
          }
    }
    __builtin_amdgcn_s_barrier();       // B: every thread has read the previous tile (right after barrier A)
    asm volatile("" ::: "memory");
    // ---- accumulators -> tile[pr][2 * bl + pc][channel]: lane pairs exchange so that each writes a (c, c+1) dword
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int ph = 0; ph < 4; ++ph) {
        const int pr = ph >> 1, pc = ph & 1;
#pragma unroll
        for (int ep = 0; ep < 2; ++ep) {
          const float v0 = acc[mt][ph][2 * ep], v1 = acc[mt][ph][2 * ep + 1];
          const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, odd ? v0 : v1), 0xB1, 0xF, 0xF, false));
          const float lo = odd ? recv : v0, hi = odd ? v1 : recv;
          const uint32_t word = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
          const int bl = mt * 16 + 4 * (lane >> 4) + 2 * ep + odd;       // dy column inside the strip
          const int ch = wid * 16 + (lane & 15) - odd;
          lds_store32u(tile0 + d2_tile_off(pr * 128 + 2 * bl + pc, ch >> 3) + (ch & 7) * 2, word);
        }
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  // ---- last tile
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  {
    const long long orow = img_out + (long long)(2 * (a0 + a.TH - 1)) * (2 * a.OW) * 64 + (long long)(2 * b0) * 64;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int idx = tid + 256 * k;
      const int pxl = idx >> 3, part = idx & 7;
      const u32x4 val = *reinterpret_cast<const u32x4*>(smem + G::NR * G::ROWB + d2_tile_off(pxl, part));
      *reinterpret_cast<u32x4*>(a.DX + orow + (long long)(pxl >> 7) * (2 * a.OW) * 64 + (pxl & 127) * 64 + part * 8) = val;
    }
  }
  if (a.mom != nullptr) {
    // the four lane groups hold the same column (a quarter of the values each): (mean, M2) per lane, then two pairwise merges
    const float quarter = 64.f * (float)a.TH;             // TH dy rows x 64 columns x 4 phases / 4 lane groups
    float mean, m2;
    shifted_to_mean_m2(ms1, ms2, mpilot, quarter, mean, m2);
    float mean_o = __shfl_xor(mean, 16, 64), m2_o = __shfl_xor(m2, 16, 64);
    chan_merge_equal(mean, m2, mean_o, m2_o, quarter);
    mean_o = __shfl_xor(mean, 32, 64);
    m2_o = __shfl_xor(m2, 32, 64);
    chan_merge_equal(mean, m2, mean_o, m2_o, 2.f * quarter);
    if (lane < 16) {
      const int col = wid * 16 + lane, slot = band * a.strips + strip;
      float* const o = a.mom + (((long long)n * 64 + col) * a.mom_slots + slot) * 2;
      o[0] = mean;
      o[1] = m2;
    }
  }
}

}  // namespace jpdse
