// bf16 weight gradient of the 3x3 stride-1 ResnetBlock convs:  dW[k][r][s][c] = sum_p dy[p][k] * x[p + (r-1, s-1)][c].
//
// wgrad_fast_kernel is bound by the L2->LDS fill (ablation: without the DMA it runs 2.2x faster), and
// per 64-pixel chunk it stages a dy tile AND an x tile for every single tap.  The three taps of one
// filter row read the SAME input row shifted by one pixel, so here a block owns a (256 k) x (128 c)
// tile of ALL THREE taps of filter row r: per chunk (64 consecutive pixels of one output row) it stages
// the dy tile once (64 px x 256 k = 32 KiB) and the input row segment once (66 px x 128 c = 16.5 KiB;
// reflect / zero padding resolved per pixel by the loader), and the B fragments of tap s are transposed
// reads (ds_read_b64_tr_b16) at pixel offset s.  Fill per tap-tile of 256 x 256: 32 KiB instead of 64 KiB.
// 8 waves = 2 (k) x 4 (c), wave tile 128 k x 32 c x 3 taps = 12 accumulator tiles (192 AGPRs); every A
// fragment feeds 3 MFMAs.  Stream-K over the (tile, chunk) space as in wgrad_fast.h.
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "wgrad_fast.h"

namespace jpdse {

struct RowWgArgs {
  const bf16_t* X;    // [N][H][W][Cs] unpadded (OH == H, OW == W: 3x3, stride 1, pad 1)
  const bf16_t* DY;   // [N][H][W][Ks]
  float* DW;          // fp32 KRSC [K][3][3][C]
  int N, H, W, Cs, C, Ks, K, reflect;
  int chunks_per_row, chunks_total, iters_per_block;
  long long total_iters;
};

template <int KS>
__device__ __forceinline__ void mma_step(uint32_t sbase, const int (&a_tr)[4], const int (&b_tr)[3], f32x16 (&acc)[3][4]) {
  constexpr int A_ROWB = 512, B_ROWB = 256;
  s16x8 af[4], bf[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) af[i] = tr_frag_asm<KS * 16 * A_ROWB, KS * 16 * A_ROWB + 4 * A_ROWB>(sbase + a_tr[i]);
#pragma unroll
  for (int s = 0; s < 3; ++s) bf[s] = tr_frag_asm<KS * 16 * B_ROWB, KS * 16 * B_ROWB + 4 * B_ROWB>(sbase + b_tr[s]);
  tr_wait(af);
  tr_wait(bf);
#pragma unroll
  for (int s = 0; s < 3; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      acc[s][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[s], acc[s][i], 0, 0, 0);
}

// ABL: timing-only ablations (wrong results): 1 = no epilogue, 2 = plain stores instead of atomics
// REFLECT: padding mode as a template parameter -- with the incremental chunk coordinates and the select-based pixel
// resolution below the loader has no branch and no division (it used to compile to ~30 scalar / exec branches and two
// divisions per chunk, executed right behind the barrier by all 8 waves).
template <int ABL, bool REFLECT>
__global__ __launch_bounds__(512) void wgrad_row_kernel(const RowWgArgs a) {
  constexpr int NW = 8, WN = 4, TM = 4;
  constexpr int BM = 256, BN = 128, BKP = 64;
  constexpr int A_ROWB = BM * 2, B_ROWB = BN * 2;           // 512, 256 bytes per pixel row
  constexpr int A_STAGE = BKP * A_ROWB;                     // 32 KiB
  constexpr int B_PIX = BKP + 2;                            // 66 input pixels
  constexpr int B_UNITS = (B_PIX + 3) / 4;                  // 17 units of 4 pixels
  constexpr int B_STAGE = B_UNITS * 1024;
  constexpr int STAGE = A_STAGE + B_STAGE;
  constexpr int A_UNITS = A_STAGE / 1024;                   // 32 units of 2 pixels
  constexpr int UNITS = A_UNITS + B_UNITS;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int c_tiles = a.Cs / BN;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const uint32_t lds0 = lds_addr_of(smem);

  // transposed fragment offsets (lane roles as in wgrad_fast.h); B: one per tap s (pixel row shifted by s)
  int a_tr[TM], b_tr[3];
  {
    const int g = lane >> 4, li = lane & 15, h = g >> 1, cb = g & 1, q = li >> 2, p = li & 3;
    const int pix = 8 * h + q;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ch = (wm * TM + i) * 32 + cb * 16 + 4 * p;
      a_tr[i] = pix * A_ROWB + ((((ch >> 3) ^ trswz<A_ROWB>(pix)) << 4) | ((ch & 7) << 1));
    }
    const int chb = wn * 32 + cb * 16 + 4 * p;
#pragma unroll
    for (int s = 0; s < 3; ++s)
      b_tr[s] = A_STAGE + (pix + s) * B_ROWB + ((((chb >> 3) ^ trswz<B_ROWB>(pix + s)) << 4) | ((chb & 7) << 1));
  }

  long long it0 = (long long)blockIdx.x * a.iters_per_block;
  long long it1 = it0 + a.iters_per_block;
  it1 = it1 < a.total_iters ? it1 : a.total_iters;
  while (it0 < it1) {
    const int tile = (int)(it0 / a.chunks_total);
    const int ch_begin = (int)(it0 - (long long)tile * a.chunks_total);
    int ch_end = ch_begin + (int)(it1 - it0);
    ch_end = ch_end < a.chunks_total ? ch_end : a.chunks_total;
    it0 += ch_end - ch_begin;
    const bool whole_tile = ch_begin == 0 && ch_end == a.chunks_total;
    const int ct = tile % c_tiles, t1 = tile / c_tiles;
    const int r = t1 % 3, kt = t1 / 3;
    const int k0 = kt * BM, c0 = ct * BN;

    f32x16 acc[3][TM];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][i][e] = 0.f;

    // Per-lane source offsets that do not depend on the chunk (the address arithmetic of the loader, not
    // the DMA itself, was what bound this kernel: ~450 VALU issue slots per chunk and wave against 48
    // MFMAs; now ~60).  A (dy): 4 units per wave, B (x): units wid, wid+8, wid+16 (< 17).
    constexpr int AU = A_UNITS / NW, BU = (B_UNITS + NW - 1) / NW;
    // the swizzle depends on (pixel & 3) only and the units of one wave are 16 (dy) / 32 (x) pixels apart: ONE lane
    // offset per operand, the unit index goes into the wave-uniform base
    const int a_pix0 = wid * 2 + (lane >> 5);
    const int a_loff0 = a_pix0 * a.Ks + k0 + (((lane & 31) ^ trswz<A_ROWB>(a_pix0)) << 3);
    const int b_q0 = wid * 4 + (lane >> 4);
    const int b_sw0 = c0 + (((lane & 15) ^ trswz<B_ROWB>(b_q0)) << 3);
    // coordinates of the next chunk to issue: (i_n, i_oh, i_cw), advanced incrementally
    int i_cw = __builtin_amdgcn_readfirstlane(ch_begin % a.chunks_per_row);    // wave-uniform: keep them in SGPRs
    int i_row = __builtin_amdgcn_readfirstlane(ch_begin / a.chunks_per_row);   // n*H + oh
    int i_oh = __builtin_amdgcn_readfirstlane(i_row % a.H);
    const int Hm1 = a.H - 1, Wm1 = a.W - 1;
    auto issue = [&](int stage) {
      char* const st = smem + stage * STAGE;
      const int ow0 = i_cw * BKP;
      int ih = i_oh + r - 1;
      bool row_ok = true;
      if constexpr (REFLECT) {
        ih = ih < 0 ? -ih : ih;
        ih = ih > Hm1 ? 2 * Hm1 - ih : ih;
      } else {
        row_ok = (unsigned)ih < (unsigned)a.H;
        ih = row_ok ? ih : 0;
      }
      const bf16_t* const dy_base = a.DY + ((long long)i_row * a.W + ow0) * a.Ks;                          // wave-uniform
      const bf16_t* const x_row = a.X + ((long long)(i_row - i_oh + ih) * a.W) * a.Cs;                    // wave-uniform
#pragma unroll
      for (int i = 0; i < AU; ++i) glds16(dy_base + (long long)(i * 2 * NW) * a.Ks + (unsigned)a_loff0, st + (wid + NW * i) * 1024);
      auto b_unit = [&](int j, int q, int sw, bool last) {
        int iw = ow0 - 1 + q;
        if constexpr (REFLECT) {
          // every lane reads a real pixel (pixels 66, 67 of the last unit are never consumed): no zero page, and the
          // address is uniform base + 32-bit lane offset
          iw = iw < 0 ? -iw : iw;
          iw = iw > Wm1 ? 2 * Wm1 - iw : iw;
          glds16(x_row + (unsigned)(__mul24(iw, a.Cs) + sw), st + (A_UNITS + wid + NW * j) * 1024);
        } else {
          bool ok = row_ok & ((unsigned)iw < (unsigned)a.W);
          if (last) ok = ok & (q < B_PIX);
          const bf16_t* const src = x_row + (unsigned)(__mul24(ok ? iw : 0, a.Cs) + sw);
          glds16(ok ? src : zero, st + (A_UNITS + wid + NW * j) * 1024);
        }
      };
      b_unit(0, b_q0, b_sw0, false);
      b_unit(1, b_q0 + 4 * NW, b_sw0, false);
      if (wid == 0) {                                 // unit 16 = pixels 64..67 (64, 65 live); lane constants rebuilt
        int l4 = lane >> 4;                           // from the lane id, in place: hoisted out of the loop it is
        asm volatile("" : "+v"(l4));                  // spilled, and the reload drains this wave's DMAs (vmcnt(0))
        const int q = 64 + l4;
        b_unit(2, q, c0 + (((lane & 15) ^ trswz<B_ROWB>(q)) << 3), true);
      }
      if (++i_cw == a.chunks_per_row) {
        i_cw = 0;
        ++i_row;
        if (++i_oh == a.H) i_oh = 0;
      }
    };

    // 3-stage ring, counted vmcnt: chunk c+1 stays in flight across the barrier while chunk c is consumed
    // and chunk c+2 is issued (the DMA latency, not its bandwidth, is what a 2-stage ring exposes).  Wave 0 issues
    // 7 pieces per chunk, the others 6; one count (6) for all: wave 0 then also waits for the oldest piece of chunk
    // c+1, issued a whole iteration earlier.
    issue(0);
    if (ch_begin + 1 < ch_end) issue(1);
    int cstage = 0, istage = 2;
    for (int c = ch_begin; c < ch_end; ++c) {
      if (c + 1 < ch_end) wait_vmcnt<UNITS / NW>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      if (c + 2 < ch_end) {
        issue(istage);
        istage = istage == 2 ? 0 : istage + 1;
      }
      const uint32_t sbase = lds0 + cstage * STAGE;
      __builtin_amdgcn_s_setprio(1);
      mma_step<0>(sbase, a_tr, b_tr, acc);
      mma_step<1>(sbase, a_tr, b_tr, acc);
      mma_step<2>(sbase, a_tr, b_tr, acc);
      mma_step<3>(sbase, a_tr, b_tr, acc);
      __builtin_amdgcn_s_setprio(0);
      cstage = cstage == 2 ? 0 : cstage + 1;
    }

    const int cc = c0 + wn * 32 + (lane & 31);
    if (cc < a.C && !(ABL & 1)) {
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int k = k0 + (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            if (k >= a.K) continue;
            float* dst = a.DW + (((long long)k * 3 + r) * 3 + s) * a.C + cc;
            if (!whole_tile && !(ABL & 2)) atomicAdd(dst, acc[s][i][e]);
            else *dst = acc[s][i][e];
          }
    }
    wait_vmcnt<0>();   // the counted waits of the next segment must see DMA operations only
    __syncthreads();   // the next segment re-uses the LDS stages
  }
}

}  // namespace jpdse
