// bf16 weight gradient of the convs fed by a THIN input (the 40-channel network inputs of G / D, and
// any input whose channel storage is not a multiple of 64) with 32 or 64 output channels:
//     dW[k][r][s][c] = sum_{n,oh,ow} dy[n,oh,ow,k] * xpad[n, oh*st + r, ow*st + s, c]
//
// In the materially padded NHWC input the S*Cs values under filter row r of output pixel ow are ONE
// contiguous run starting at pixel ow*st, and the runs of neighbouring output pixels overlap.  The
// generic / run-mode kernels stage every pixel's run separately (S-fold redundant L2->LDS traffic,
// the resource these kernels are bound by).  Here a block owns one filter row r and walks over
// strips of 64 output pixels of one output row: the strip's input pixels are copied ONCE, densely
// ([pixel][Cs], no padding, no swizzle) by LDS-DMA, and the GEMM B operand
//     B[pix][j] = lds[pix * st*Cs + j],   j in [0, S*Cs)  (column j <-> tap s = j / Cs, channel j % Cs)
// is formed by the per-lane addresses of ds_read_b64_tr_b16 (overlapping rows cost nothing).  dy is
// staged as in wgrad_fast.h.  M = Ks (32*TM) output channels, N = S*Cs run columns in NW*NT 32-wide
// tiles (wave w owns NT of them), reduction over the strip's 64 pixels per step; 2 LDS stages.
// Blocks accumulate their strips in registers and store the result into their pixel range's fp32 slab;
// slab_reduce_kernel adds the slabs in a fixed order (round 1 used fp32 atomics into the zeroed gradient).
//
// RR > 1: one block handles RR filter rows (run strips of RR input rows next to ONE dy strip).
// `transposed`: the roles are swapped for convs with <= 8 OUTPUT channels (the 64->3 / 32->3 heads):
// the "dy" operand is the padded conv input (M = its 32/64 channels) and the run operand is dy,
// zero-padded by (R-1, S-1), 8 channels per pixel: G[c][r'][s'*8 + k] = sum_p xpad[p][c] *
// dypad[p + (r', s')][k] is the gradient of tap (R-1-r', S-1-s').  All R rows share the x strip
// (RR = R), so the input is staged once instead of R times.
#pragma once
#include "common.h"
#include "gemm_fast.h"
#include "wgrad_fast.h"

namespace jpdse {

struct ThinWgArgs {
  const bf16_t* XP;    // padded input [N][Hp][Wp][Cs]
  const bf16_t* DY;    // [N][OH][OW][Ks]
  float* DW;           // fp32 KRSC (written by slab_reduce_kernel)
  float* partial;      // [ranges][slab_stride] fp32: pixel range p stores its partial gradient into slab p (no atomics)
  long long slab_stride;
  int N, OH, OW, Hp, Wp, Cs, C, Ks, K, R, S, st;
  int chunks_per_row, strips_total, strips_per_block;
  int ranges, row_groups;   // grid = roundup(ranges, 8) * row_groups blocks, see launch_wgrad_thin_pitch
  int x_units;         // 1 KiB DMA units per input strip
  long long x_limit;   // elements of XP that may be read (tensor + zeroed slack)
  int transposed;
  // In-loader padding (no materialised padded copies): XP / DY are the UNPADDED tensors, the padded geometry
  // (Hp, Wp for the run operand; OH, OW for the dy-side operand) is virtual and every lane resolves its pixel.
  int unpadded;        // 1: the fields below are valid
  int RH, RW, r_pad, r_reflect;   // run operand: real dims, padding that produced Hp x Wp
  int AH, AW, a_pad, a_reflect;   // dy-side operand: real dims, padding that produced OH x OW (0 for a true dy)
};

// TM m-tiles per wave, WM x WN waves (M = 32*TM*WM channels of the dy operand, WN*NT run tiles)
// PITCH: st*Cs when known at compile time (the fragment offsets of the run operand become immediates: 24 fewer
// live address registers, one more wave per SIMD), 0 = runtime
template <int TM, int WM, int WN, int NT, int RR, int PITCH, bool TRANSPOSED>
__global__ __launch_bounds__(64 * WM * WN) void wgrad_thin_kernel(const ThinWgArgs a) {
  constexpr int NW = WM * WN;
  constexpr int A_ROWB = TM * WM * 64;            // bytes per dy pixel row (Ks = 32*TM*WM channels)
  constexpr int A_STAGE = 64 * A_ROWB;
  constexpr int A_UNITS = A_STAGE / 1024;
  constexpr int A_PPU = 1024 / A_ROWB;            // pixels per DMA unit
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int stage_bytes = A_STAGE + RR * a.x_units * 1024;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const uint32_t lds0 = lds_addr_of(smem);
  // 8 * row_groups consecutive blocks = 8 pixel ranges x all filter-row groups; block b and b + 8 (same XCD) = the
  // same pixel range, next row group
  const int grp = blockIdx.x / (8 * a.row_groups), within = blockIdx.x - grp * (8 * a.row_groups);
  const int range = grp * 8 + (within & 7);
  if (range >= a.ranges) return;
  const int r0 = (within >> 3) * RR;
  const int s_begin = range * a.strips_per_block;
  int s_end = s_begin + a.strips_per_block;
  s_end = s_end < a.strips_total ? s_end : a.strips_total;
  if (s_begin >= s_end) return;
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);
  const int pitch = PITCH ? PITCH : a.st * a.Cs;  // elements between consecutive output pixels' runs
  const int units = A_UNITS + RR * a.x_units;

  // transposed fragment offsets (lane roles as in wgrad_fast.h)
  int a_tr[TM], b_tr[NT];
  {
    const int g = lane >> 4, li = lane & 15, h = g >> 1, cb = g & 1, q = li >> 2, p = li & 3;
    const int pix = 8 * h + q;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ch = (wm * TM + i) * 32 + cb * 16 + 4 * p;
      a_tr[i] = pix * A_ROWB + ((((ch >> 3) ^ trswz<A_ROWB>(pix)) << 4) | ((ch & 7) << 1));
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
      b_tr[j] = A_STAGE + (pix * pitch + (wn * NT + j) * 32 + cb * 16 + 4 * p) * 2;
  }
  const int pb16 = 16 * pitch * 2, pb4 = 4 * pitch * 2;
  const int xs_bytes = a.x_units * 1024;

  f32x16 acc[RR][TM][NT];
#pragma unroll
  for (int rr = 0; rr < RR; ++rr)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[rr][i][j][e] = 0.f;

  // One 16-pixel k-step = TM + RR*NT transposed fragments (asm reads, see tr_frag_asm) and TM*RR*NT MFMAs.
  // The fragments are double buffered: the reads of k-step KS+1 are issued right after the wait for
  // k-step KS and fly while its MFMAs execute (a wait after every read group exposed the LDS latency 8
  // times per strip: 0.60 -> 0.88 ms on the first conv).
  auto rd_frags = [&]<int KS>(uint32_t sbase, s16x8 (&af)[TM], s16x8 (&bf)[RR * NT]) {
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = tr_frag_asm<KS * 16 * A_ROWB, KS * 16 * A_ROWB + 4 * A_ROWB>(sbase + a_tr[i]);
#pragma unroll
    for (int rr = 0; rr < RR; ++rr)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if constexpr (PITCH != 0) {
          bf[rr * NT + j] = tr_frag_asm<KS * 32 * PITCH, KS * 32 * PITCH + 8 * PITCH>(sbase + rr * xs_bytes + b_tr[j]);
        } else {
          const uint32_t lo = sbase + rr * xs_bytes + b_tr[j] + KS * pb16;
          bf[rr * NT + j] = tr_frag_asm2(lo, lo + pb4);
        }
      }
  };
  auto mma_frags = [&](const s16x8 (&af)[TM], const s16x8 (&bf)[RR * NT]) {
#pragma unroll
    for (int rr = 0; rr < RR; ++rr)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[rr][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[rr * NT + j], acc[rr][i][j], 0, 0, 0);
  };
  auto wait_frags = [&](s16x8 (&af)[TM], s16x8 (&bf)[RR * NT]) {
    tr_wait(af);
    tr_wait(bf);
  };

  // lane constants of the loader (the per-unit part is wave-uniform and stays on the scalar unit)
  const int a_pl = lane / (A_ROWB / 16);                      // pixel within a dy unit
  const int a_slot_off = ((lane % (A_ROWB / 16)) ^ trswz<A_ROWB>(a_pl)) << 3;
  const int a_lane_off = a_pl * a.Ks + a_slot_off;
  constexpr int CS = PITCH == 80 ? 40 : PITCH;        // channel storage when the pitch is a compile-time constant
  static_assert(A_PPU % 4 == 0 || A_ROWB >= 256, "swizzle must not depend on the unit index");
  // Strips are walked DOWN the image: strip index = (n * chunks_per_row + chunk column) * OH + oh.  The block of filter
  // row r reads input row oh*st + r, which the block of row r - st (same pixel range, same XCD, see the launcher) reads
  // one strip later: with the row-major order of round 1 that reuse was a whole image row of strips apart, further than
  // the XCD's L2 reaches, and every filter row re-fetched the input (7x on the first 7x7 conv).
  // Coordinates advance incrementally (wave-uniform, no division per strip).
  int i_oh = __builtin_amdgcn_readfirstlane(s_begin % a.OH);
  int i_chunk = __builtin_amdgcn_readfirstlane((s_begin / a.OH) % a.chunks_per_row);
  int i_n = __builtin_amdgcn_readfirstlane((s_begin / a.OH) / a.chunks_per_row);
  int i_row = __builtin_amdgcn_readfirstlane(i_n * a.OH + i_oh);                  // n*OH + oh
  auto issue = [&](int /*strip*/, int stage) {
    char* const st = smem + stage * stage_bytes;
    const int row = i_row, oh = i_oh, n = i_n;
    const int ow0 = i_chunk * 64;
    ++i_row;
    if (++i_oh == a.OH) {
      i_oh = 0;
      i_row -= a.OH;
      if (++i_chunk == a.chunks_per_row) { i_chunk = 0; ++i_n; i_row += a.OH; }
    }
    const bf16_t* const dy_base = a.DY + ((long long)row * a.OW + ow0) * a.Ks;                       // uniform
    const long long x_base = (((long long)n * a.Hp + oh * a.st + r0) * a.Wp + (long long)ow0 * a.st) * a.Cs;
    const long long x_row = (long long)a.Wp * a.Cs;
    const int px_left = a.OW - ow0;                  // valid output pixels from ow0 on
    // Two flat passes (dy-side units, then run units), the padded / in-loader-padding variants split ONCE per strip:
    // the single loop with nested if / else per unit compiled to ~15 scalar and exec branches per DMA.
    constexpr int A_ITERS = (A_UNITS + NW - 1) / NW;
    if (!a.unpadded) {
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        const int u = wid + it * NW;
        if (A_UNITS % NW != 0 && u >= A_UNITS) break;               // wave-uniform
        const int pix0 = u * A_PPU;
        const bool ok = pix0 + a_pl < px_left;
        const bf16_t* const src = dy_base + (ok ? pix0 * a.Ks + a_lane_off : 0);
        glds16(ok ? src : zero, st + u * 1024);
      }
      for (int ub = wid; ub < RR * a.x_units; ub += NW) {
        int rr = 0, uu = ub;
        if constexpr (RR > 1) {       // wave-uniform; RR == 1 keeps the division out of the loader
          rr = ub / a.x_units;
          uu = ub - rr * a.x_units;
        }
        const long long e0 = x_base + rr * x_row + (long long)uu * 512;   // uniform: first element of the unit
        const bool ok = (r0 + rr < a.R) & (e0 + lane * 8 + 8 <= a.x_limit);
        const bf16_t* const src = a.XP + (ok ? e0 + lane * 8 : 0);
        glds16(ok ? src : zero, st + (A_UNITS + ub) * 1024);
      }
    } else {
#pragma unroll
      for (int it = 0; it < A_ITERS; ++it) {
        const int u = wid + it * NW;
        if (A_UNITS % NW != 0 && u >= A_UNITS) break;               // wave-uniform
        const int pix0 = u * A_PPU;
        // virtual (oh, ow0 + pix) of the padded grid -> pixel of the real tensor
        int ay = oh - a.a_pad, ax = ow0 + pix0 + a_pl - a.a_pad;
        if (a.a_reflect) {
          ay = ay < 0 ? -ay : (ay >= a.AH ? 2 * (a.AH - 1) - ay : ay);
          ax = ax < 0 ? -ax : (ax >= a.AW ? 2 * (a.AW - 1) - ax : ax);
        }
        const bool arow_ok = (unsigned)ay < (unsigned)a.AH;                                      // uniform
        const bf16_t* const arow = a.DY + ((long long)n * a.AH + (arow_ok ? ay : 0)) * a.AW * a.Ks;
        const bool ok = arow_ok & (pix0 + a_pl < px_left) & ((unsigned)ax < (unsigned)a.AW);
        const bf16_t* const src = arow + (ok ? __mul24(ax, a.Ks) + a_slot_off : 0);
        glds16(ok ? src : zero, st + u * 1024);
      }
      for (int ub = wid; ub < RR * a.x_units; ub += NW) {
        int rr = 0, uu = ub;
        if constexpr (RR > 1) {
          rr = ub / a.x_units;
          uu = ub - rr * a.x_units;
        }
        // element offset inside the virtual padded row -> (pixel, channel slot) -> real pixel
        const int el = uu * 512 + lane * 8;
        const int pxo = PITCH ? el / CS : el / a.Cs;
        const int ch = el - pxo * (PITCH ? CS : a.Cs);
        int ry = oh * a.st + r0 + rr - a.r_pad, rx = ow0 * a.st + pxo - a.r_pad;
        if (a.r_reflect) {
          ry = ry < 0 ? -ry : (ry >= a.RH ? 2 * (a.RH - 1) - ry : ry);
          rx = rx < 0 ? -rx : (rx >= a.RW ? 2 * (a.RW - 1) - rx : rx);
        }
        const bool rrow_ok = (r0 + rr < a.R) & (oh * a.st + r0 + rr < a.Hp) & ((unsigned)ry < (unsigned)a.RH);   // uniform
        const bf16_t* const rrow = a.XP + ((long long)n * a.RH + (rrow_ok ? ry : 0)) * a.RW * (PITCH ? CS : a.Cs);
        const bool ok = rrow_ok & (ow0 * a.st + pxo < a.Wp) & ((unsigned)rx < (unsigned)a.RW);
        const bf16_t* const src = rrow + (ok ? __mul24(rx, PITCH ? CS : a.Cs) + ch : 0);
        glds16(ok ? src : zero, st + (A_UNITS + ub) * 1024);
      }
    }
  };

  int stage = 0;
  issue(s_begin, 0);
  wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  for (int sidx = s_begin; sidx < s_end; ++sidx) {
    if (sidx + 1 < s_end) issue(sidx + 1, stage ^ 1);
    const uint32_t sbase = lds0 + stage * stage_bytes;
    __builtin_amdgcn_s_setprio(1);
    {
      s16x8 afA[TM], bfA[RR * NT], afB[TM], bfB[RR * NT];
      rd_frags.template operator()<0>(sbase, afA, bfA);
      wait_frags(afA, bfA);
      rd_frags.template operator()<1>(sbase, afB, bfB);
      mma_frags(afA, bfA);
      wait_frags(afB, bfB);
      rd_frags.template operator()<2>(sbase, afA, bfA);
      mma_frags(afB, bfB);
      wait_frags(afA, bfA);
      rd_frags.template operator()<3>(sbase, afB, bfB);
      mma_frags(afA, bfA);
      wait_frags(afB, bfB);
      mma_frags(afB, bfB);
    }
    __builtin_amdgcn_s_setprio(0);
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    stage ^= 1;
  }

  const int run = a.S * a.Cs;
  float* const out = a.partial + (long long)range * a.slab_stride;
  if constexpr (TRANSPOSED) {
    // The gradient row of one (output channel c, tap) is the M = a.K contiguous floats of the conv's input
    // channels: transpose each filter row's tile through LDS so that one store instruction covers a
    // contiguous row instead of 64 scattered cache lines.
    constexpr int MROWS = 32 * TM * WM, COLS = 32 * WN * NT, TPITCH = MROWS + 1;
    float* const tbuf = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int rr = 0; rr < RR; ++rr) {
      const int r = r0 + rr;
      __syncthreads();
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = (wn * NT + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 16; ++e) {
            const int k = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            tbuf[col * TPITCH + k] = acc[rr][i][j][e];
          }
      }
      __syncthreads();
      if (r >= a.R) continue;
      for (int idx = tid; idx < a.S * a.C * MROWS; idx += 64 * NW) {
        const int k = idx % MROWS, sc = idx / MROWS;
        const int s_out = sc / a.C, c = sc - s_out * a.C;
        if (k >= a.K) continue;
        out[(((long long)c * a.R + (a.R - 1 - r)) * a.S + (a.S - 1 - s_out)) * a.K + k] = tbuf[(s_out * a.Cs + c) * TPITCH + k];
      }
    }
    return;
  }
#pragma unroll
  for (int rr = 0; rr < RR; ++rr) {
    const int r = r0 + rr;
    if (r >= a.R) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int col = (wn * NT + j) * 32 + (lane & 31);
      if (col >= run) continue;
      const int s_out = col / a.Cs, c = col - s_out * a.Cs;
      if (c >= a.C) continue;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int k = (wm * TM + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
          if (k >= a.K) continue;
          out[(((long long)k * a.R + r) * a.S + s_out) * a.C + c] = acc[rr][i][j][e];
        }
      }
    }
  }
}

}  // namespace jpdse
