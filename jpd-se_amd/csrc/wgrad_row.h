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
template <int ABL>
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
    int a_loff[AU], b_q[BU], b_sw[BU];
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      const int pix = (wid + NW * i) * 2 + (lane >> 5);
      a_loff[i] = pix * a.Ks + k0 + (((lane & 31) ^ trswz<A_ROWB>(pix)) << 3);
    }
#pragma unroll
    for (int j = 0; j < BU; ++j) {
      const int q = (wid + NW * j) * 4 + (lane >> 4);
      b_q[j] = q;
      b_sw[j] = c0 + (((lane & 15) ^ trswz<B_ROWB>(q)) << 3);
    }
    auto issue = [&](int chunk, int stage) {
      char* const st = smem + stage * STAGE;
      const int cw = chunk % a.chunks_per_row;
      const int row = chunk / a.chunks_per_row;       // n*H + oh
      const int oh = row % a.H, n = row / a.H;
      const int ow0 = cw * BKP;
      int ih = oh + r - 1;
      bool row_ok = true;
      if (a.reflect) ih = ih < 0 ? -ih : (ih >= a.H ? 2 * (a.H - 1) - ih : ih);
      else row_ok = (unsigned)ih < (unsigned)a.H;
      const bf16_t* const dy_base = a.DY + ((long long)row * a.W + ow0) * a.Ks;        // wave-uniform
      const bf16_t* const x_row = a.X + (((long long)n * a.H + ih) * a.W) * a.Cs;       // wave-uniform
#pragma unroll
      for (int i = 0; i < AU; ++i) glds16(dy_base + a_loff[i], st + (wid + NW * i) * 1024);
#pragma unroll
      for (int j = 0; j < BU; ++j) {
        if (wid + NW * j < B_UNITS) {
          int iw = ow0 - 1 + b_q[j];
          bool ok = row_ok && b_q[j] < B_PIX;
          if (a.reflect) iw = iw < 0 ? -iw : (iw >= a.W ? 2 * (a.W - 1) - iw : iw);
          else ok = ok && (unsigned)iw < (unsigned)a.W;
          const bf16_t* src = ok ? x_row + (__mul24(iw, a.Cs) + b_sw[j]) : zero;
          glds16(src, st + (A_UNITS + wid + NW * j) * 1024);
        }
      }
    };

    // 3-stage ring, counted vmcnt: chunk c+1 stays in flight across the barrier while chunk c is consumed
    // and chunk c+2 is issued (the DMA latency, not its bandwidth, is what a 2-stage ring exposes)
    issue(ch_begin, 0);
    if (ch_begin + 1 < ch_end) issue(ch_begin + 1, 1);
    int cstage = 0, istage = 2;
    for (int c = ch_begin; c < ch_end; ++c) {
      if (c + 1 < ch_end) {
        if (wid == 0) wait_vmcnt<(UNITS + NW - 1) / NW>(); else wait_vmcnt<UNITS / NW>();
      } else {
        wait_vmcnt<0>();
      }
      __builtin_amdgcn_s_barrier();
      if (c + 2 < ch_end) {
        issue(c + 2, istage);
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
