// Persistent, cross-tile-pipelined form of the fast implicit-GEMM kernel for the SHORT-K layers (round 4):
// PatchGAN layers 1-2 (4x4 stride 2, K = 1024 / 2048: 16 / 32 K-tiles), the 3x3 stride-2 down-sampling convs of the generator
// (18 / 36 K-tiles), their second-scale and ConvTranspose-gradient siblings.
//
// What bounds these layers on gemm_fast_kernel (measured): the L2 -> LDS fill of a CU.  A 64 x 128 (128 x 128)
// tile stages 24 (32) KiB per 64-deep K-tile for 1.05 (2.1) MFLOP; at the 500-600 TFLOP/s these layers reach that is ~50 GB/s per
// CU, three quarters of the 66-73 GB/s a CU gathers out of L2 (MI355X_MICROARCH.md, "Indexed rows").  Small tiles were chosen
// because two or three co-resident blocks hide each other's prologue and epilogue -- at twice the fill bytes per FLOP of the
// 256 x 128 tile.  The way out is the big tile WITHOUT its exposed prologue / epilogue: ONE workgroup per CU walks a contiguous
// range of 256 x 128 tiles and the K-tile stream never stops at a tile boundary -- the loader cursor runs ahead INTO THE NEXT TILE
// while the current tile's last K-steps and its epilogue execute (two K-tiles = 96 KiB in flight), so a tile's first fragments
// are in LDS when its first MFMA issues.  The epilogue is staged through the ring slot the tile's last K-step has just
// consumed (free until the issue of the NEXT step re-targets it): one extra barrier per tile, no LDS beyond the 3-stage ring.
//
// First version, measured and replaced (profiles/r04_pers_v1_ab.txt): 128 x 128 tiles, 4 waves (one per SIMD), 4-stage ring with
// private per-wave staging -- 400-500 TFLOP/s against gemm_fast's 550-620 on the same layers: the same fill bytes per FLOP as the
// co-resident 128-row configuration, and with one wave per SIMD the loader's address arithmetic (re-resolved every K-tile for
// 64-channel inputs) runs with the matrix pipe idle.
//
// Structure: 8 waves (two per SIMD: one wave's loader VALU overlaps its partner's MFMAs), 256 x 128 (256 x 64) tile, wave tile
// 64 x 64 (64 x 32), BK = 64, LDS-DMA staging with the fast kernel's swizzle, one raw s_barrier per K-step, counted vmcnt.
// The counts: at step g the stage needed was issued STAGES-1 = 2 issues ago; behind it are one more stage (LW DMA instructions)
// and, right after an epilogue, that epilogue's stores.  The wait is vmcnt(LW) in both cases -- with stores in the window that
// over-waits (the epilogue has given the DMA time to land anyway), and it is correct whether or not stores retire in order with
// loads.  Tail of the block (loader out of tiles): the count shrinks with the stages actually outstanding.
// Every LDS access of the epilogue is inline asm (an ordinary LDS store or load issued while an LDS-DMA is in flight gets
// `s_waitcnt vmcnt(0)` from hipcc).  No split-K, no bias, no fused addend / mask operands (an ordinary global load inside the DMA
// span makes hipcc drain the ring: cdna_hip_programming.md, "While a glds is in flight"); the launcher keeps such problems on
// gemm_fast_kernel.
#pragma once
#include "gemm_fast.h"
#include "gemm_halo.h"
#include "head_fwd.h"
#include "conv_rows.h"

namespace jpdse {

template <int TN>
__global__ __launch_bounds__(512) void gemm_pers_kernel(const FastBatch batch, const int total_tiles) {
  constexpr int NW = 8, WN = 2, TM = 2, STAGES = 3;
  constexpr int BM = 256, BN = WN * TN * 32;
  constexpr int A_TILE = BM * 128, B_TILE = BN * 128, STAGE_BYTES = A_TILE + B_TILE;
  constexpr int AU = BM / 8 / NW, BU = BN / 8 / NW, LW = AU + BU;     // 1 KiB DMA units per wave and K-tile
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "every wave stages the same number of units");
  constexpr int EP_PITCH = TN * 64 + 16;                               // staged row: TN * 32 channels (bf16) + pad
  constexpr int EP_WAVE = 32 * EP_PITCH + 64 * 4;                      // 32 staged rows + the wave's 64 row offsets (32-bit)
  static_assert(NW * EP_WAVE <= STAGE_BYTES, "the eight waves' staging regions fit one ring slot");
  constexpr int VPRW = TN * 4;                                         // 16-byte vectors per staged row
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int lrow = lane >> 3, lslot = lane & 7;
  const uint32_t smem0 = lds_addr32(smem);
  const bf16_t* const zero = reinterpret_cast<const bf16_t*>(g_zero_page);

  // this block's contiguous tile range
  const int nblk = (int)gridDim.x, bid = (int)blockIdx.x;
  const int tile_begin = (int)((long long)total_tiles * bid / nblk), tile_end = (int)((long long)total_tiles * (bid + 1) / nblk);
  if (tile_begin >= tile_end) return;

  auto problem_of = [&](int tile) {
    int q = 0;
#pragma unroll
    for (int i = 1; i < 4; ++i)
      if (i < batch.n && tile >= batch.first_tile[i]) q = i;
    return q;
  };

  // fragment read offsets (constant for the whole launch)
  int a_rd[TM][4], b_rd[TN][4];
  {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = wm * TM * 32 + i * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a_rd[i][ks] = swz128(row, 2 * ks + h);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int row = wn * TN * 32 + j * 32 + r;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) b_rd[j][ks] = A_TILE + swz128(row, 2 * ks + h);
    }
  }

  // ---- loader state: the cursor tile (runs ahead of the consumer tile) ----------------------------------------------------
  int l_tile = tile_begin, l_t = 0, l_T = 0;
  int ic = 0, is = 0, ir = 0, l_CC = 1, l_S = 1;
  int l_IH = 1, l_IW = 1, l_Cs = 64, l_xsh = 0, l_btr = 0, l_bts = 0;
  bool l_refl = false;
  const bf16_t* a_xb[AU];
  int a_oh[AU], a_ow[AU], a_step[AU];
  const bf16_t* a_src[AU];
  const bf16_t* b_ptr[BU];
  int istage = 0, n_issued = 0;

  auto setup_tile = [&](int tile) {
    const int q = problem_of(tile);
    const FastArgs& a = batch.p[q];
    const int tile_id = tile - batch.first_tile[q];
    const int tiles_n = (a.Ks + BN - 1) / BN;
    const int tile_m = tile_id / tiles_n, tile_n = tile_id - tile_m * tiles_n;      // the N-tiles of an M-tile are consecutive
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    l_CC = a.Cs >> 6;
    l_S = a.S;
    l_T = a.R * a.S * l_CC;
    l_t = 0;
    ic = is = ir = 0;
    l_IH = a.IH; l_IW = a.IW; l_Cs = a.Cs;
    l_xsh = (int)(a.x_sh ? a.x_sh : (long long)a.IW * a.Cs);
    l_refl = a.reflect != 0;
    l_btr = a.b_tap_r ? a.b_tap_r : a.S * a.Cs;
    l_bts = a.b_tap_s ? a.b_tap_s : a.Cs;
    const long long x_sn = a.x_sn ? a.x_sn : (long long)a.IH * a.IW * a.Cs;
    const long long ktot = a.b_stride ? a.b_stride : (long long)a.R * a.S * a.Cs;
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      const int row = (wid + i * NW) * 8 + lrow;
      int m = m0 + row;
      m = m < a.M ? m : a.M - 1;
      const int ow = m % a.OW, t = m / a.OW, oh = t % a.OH, n = t / a.OH;
      a_xb[i] = a.X + (long long)n * x_sn + ((lslot ^ (row >> 1)) & 7) * 8;
      a_oh[i] = oh * a.sy - a.py;
      a_ow[i] = ow * a.sx - a.px;
    }
#pragma unroll
    for (int j = 0; j < BU; ++j) {
      const int row = (wid + j * NW) * 8 + lrow;
      int br = n0 + row;
      br = br < a.b_rows ? br : a.b_rows - 1;
      b_ptr[j] = a.B + (long long)br * ktot + ((lslot ^ (row >> 1)) & 7) * 8;
    }
  };
  auto retap = [&]() {
    const int IHm1 = l_IH - 1, IWm1 = l_IW - 1;
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      const int ih = a_oh[i] + ir, iw = a_ow[i] + is;
      int rh = ih < 0 ? -ih : ih, rw = iw < 0 ? -iw : iw;
      rh = rh > IHm1 ? 2 * IHm1 - rh : rh;
      rw = rw > IWm1 ? 2 * IWm1 - rw : rw;
      rh = rh < 0 ? 0 : rh;
      rw = rw < 0 ? 0 : rw;
      const bool ok = l_refl | (((unsigned)ih < (unsigned)l_IH) & ((unsigned)iw < (unsigned)l_IW));
      const bf16_t* const src = a_xb[i] + (unsigned)(rh * l_xsh + rw * l_Cs);
      a_src[i] = ok ? src : zero;
      a_step[i] = ok ? 64 : 0;
    }
  };
  auto issue = [&]() {                                  // one K-tile of the cursor tile into ring slot `istage`
    char* const st = smem + istage * STAGE_BYTES;
    if (ic == 0) retap();
#pragma unroll
    for (int i = 0; i < AU; ++i) {
      glds16(a_src[i], st + (wid + i * NW) * 1024);
      a_src[i] += a_step[i];
    }
    const long long koff = (long long)ir * l_btr + is * l_bts + ic * 64;
#pragma unroll
    for (int j = 0; j < BU; ++j) glds16(b_ptr[j] + koff, st + A_TILE + (wid + j * NW) * 1024);
    if (++ic == l_CC) {
      ic = 0;
      if (++is == l_S) { is = 0; ++ir; }
    }
    istage = istage == STAGES - 1 ? 0 : istage + 1;
    ++n_issued;
    if (++l_t == l_T) {                                 // the cursor moves on to the next tile of this block
      ++l_tile;
      if (l_tile < tile_end) setup_tile(l_tile);
    }
  };

  setup_tile(tile_begin);
#pragma unroll 1
  for (int i = 0; i < STAGES - 1; ++i)
    if (l_tile < tile_end) issue();

  int g = 0, cstage = 0;                                // consumer step (= index of the stage it needs), its ring slot
  for (int tile = tile_begin; tile < tile_end; ++tile) {
    const int q = problem_of(tile);
    const FastArgs& a = batch.p[q];
    const int T_total = a.R * a.S * (a.Cs >> 6);
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    for (int t = 0; t < T_total; ++t, ++g) {
      const int ahead = n_issued - 1 - g;               // stages issued behind the one this step consumes (wave-uniform)
      if (ahead >= STAGES - 2) wait_vmcnt<(STAGES - 2) * LW>();
      else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      const char* const st = smem + cstage * STAGE_BYTES;
      if (l_tile < tile_end) issue();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        s16x8 af[TM], bf[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const s16x8*>(st + a_rd[i][ks]);
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const s16x8*>(st + b_rd[j][ks]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      __builtin_amdgcn_s_setprio(0);
      cstage = cstage == STAGES - 1 ? 0 : cstage + 1;
    }

    // ---- per-wave epilogue: rows [wm * 64, +64) x channels [wn * TN * 32, + TN * 32) of this tile ---------------------------
    // Every LDS access here is inline asm: an ordinary LDS store or load issued while an LDS-DMA is in flight gets
    // `s_waitcnt vmcnt(0)` from hipcc (it cannot prove that the two do not overlap) -- that would drain the ring at every tile.
    // The accumulators pass through VALU (bias, activation, bf16 pack) on their way, so the MFMA -> VALU wait states apply.
    // staging region: the ring slot the last K-step consumed (cstage - 1); every wave has finished reading it behind this barrier,
    // and the DMA that re-targets it is issued behind the barrier of the NEXT K-step
    __builtin_amdgcn_s_barrier();
    const uint32_t ep_tile = smem0 + (cstage == 0 ? STAGES - 1 : cstage - 1) * STAGE_BYTES + wid * EP_WAVE;   // 32 staged rows ...
    const uint32_t ep_rows = ep_tile + 32 * EP_PITCH;                                                          // ... and 64 row offsets
    const int tile_id = tile - batch.first_tile[q];
    const int tiles_n = (a.Ks + BN - 1) / BN;
    const int tile_m = tile_id / tiles_n, tile_n = tile_id - tile_m * tiles_n;
    const int m0 = tile_m * BM, n0w = tile_n * BN + wn * TN * 32;
    {
      const int m = m0 + wm * 64 + lane;
      uint32_t off = 0xffffffffu;                       // element offsets fit 31 bits (launcher-checked)
      if (m < a.M) {
        const int ow = m % a.OW, tt = m / a.OW, oh = tt % a.OH, n = tt / a.OH;
        off = (uint32_t)(a.out_base + n * a.out_sn + oh * a.out_sh + ow * a.out_sw);
      }
      lds_store32u(ep_rows + lane * 4, off);
    }
    const int odd = lane & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int lcol = j * 32 + (lane & 31);
        const int col = n0w + lcol;
        const bool live = col < a.Kout;
        // no bias here (a global load inside the DMA span drains the ring; the launcher keeps biased layers on gemm_fast_kernel);
        // activation none / ReLU / LeakyReLU as a factor for the negative values
        const float nslope = a.act == JPDSE_ACT_RELU ? 0.f : (a.act == JPDSE_ACT_LRELU ? a.slope : 1.f);
#pragma unroll
        for (int ep2 = 0; ep2 < 8; ++ep2) {
          const int e = 2 * ep2;
          const float r0 = acc[i][j][e], r1 = acc[i][j][e + 1];
          const float v0 = live ? (r0 > 0.f ? r0 : r0 * nslope) : 0.f;
          const float v1 = live ? (r1 > 0.f ? r1 : r1 * nslope) : 0.f;
          const float recv = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, odd ? v0 : v1), 0xB1, 0xF, 0xF, false));
          const float lo = odd ? recv : v0, hi = odd ? v1 : recv;
          const uint32_t word = (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
          const int row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) + odd;
          lds_store32u(ep_tile + row * EP_PITCH + (lcol - odd) * 2, word);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < 32 * VPRW / 64; ++k) {
        const int idx = lane + 64 * k;
        const int row = idx / VPRW, v = idx - row * VPRW;
        uint32_t off;
        asm volatile("ds_read_b32 %0, %1" : "=v"(off) : "v"(ep_rows + (i * 32 + row) * 4));
        s16x8 val = lds_read128_asm(ep_tile + row * EP_PITCH + v * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(off), "+v"(val));
        if (off != 0xffffffffu && n0w + v * 8 < a.Ks) *reinterpret_cast<s16x8*>(a.Y + off + n0w + v * 8) = val;
      }
    }
  }
}

}  // namespace jpdse
